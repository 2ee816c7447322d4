-- A robust (p-norm) smoothness term through lib.t's L_p: the difference is weighted by sqrt((|d| + eps)^(p-2)) held CONSTANT -- value only, no
-- derivative (IRLS) -- so the Jacobian of a Gauss-Newton step is the weighted difference operator.  (Two channels: lib.t's L_2_norm of a SCALAR is
-- the scalar itself, sign included.)
N = Dims("N")
Inputs {
    S = Unknown(float2, {N}, 0),
    A = Array(float2, {N}, 1),
    pNorm = Param(float, 2)
}
x = N()
w_reg = 1.5
r = Residuals {
    fit = S(x) - A(x),
    reg = w_reg * Select(InBounds(x + 1), L_p(S(x) - S(x + 1), pNorm, {x}), 0)
}
