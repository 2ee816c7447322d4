-- One gain per image row: the residual runs over (x, y), its unknown G(y) does not vary with the innermost iteration dimension x, so all lanes of a
-- wave scatter into the same one or two unknowns (the case the reference handles with get_peers / reduce_peers, thallo.t:3349-3399).
W, H = Dims("W", "H")
Inputs {
    I = Array(float, {W, H}, 0),
    T = Array(float, {W, H}, 1),
    G = Unknown(float, {H}, 2)
}
UsePreconditioner(true)
x, y = W(), H()
w_reg = 0.1
r = Residuals {
    fit = I(x, y) * G(y) - T(x, y),
    reg = w_reg * (G(y) - 1.0)
}
