-- test energy for Sparse maps over a 2-D domain: every pixel is tied to one other pixel of its row (Right) and one of its column (Down),
-- picked by the caller; a masked data term keeps the problem determined.
local W, H = Dims("W", "H")
Inputs {
    w_fit = Param(float, 0),
    X     = Unknown(float, {W, H}, 1),
    Data  = Array(float, {W, H}, 2),
    Mask  = Array(float, {W, H}, 3),
    Right = Sparse({W, H}, {W}, 4),
    Down  = Sparse({W, H}, {H}, 5)
}
local x, y = W(), H()
Residuals {
    fit   = w_fit * Mask(x, y) * (X(x, y) - Data(x, y)),
    tie_x = 0.5 * (X(x, y) - X(Right(x, y), y)),
    tie_y = X(x, y) - X(x, Down(x, y))
}
