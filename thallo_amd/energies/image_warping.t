-- As-rigid-as-possible 2D image warping (problem specification for libThallo).
-- Unknowns: per-pixel warped position `Offset` and local rotation `Angle`.
-- Same energy as the reference's examples/image_warping/image_warping.t; written for this repo.
local W, H = Dims("W", "H")
Inputs {
    Offset      = Unknown(thallo_float2, {W, H}, 0),
    Angle       = Unknown(thallo_float,  {W, H}, 1),
    UrShape     = Array(thallo_float2,   {W, H}, 2),   -- rest-pose pixel positions
    Constraints = Array(thallo_float2,   {W, H}, 3),   -- target positions, negative = unconstrained
    Mask        = Array(thallo_float,    {W, H}, 4),   -- non-zero = pixel outside the mesh
    w_fitSqrt   = Param(float, 5),
    w_regSqrt   = Param(float, 6)
}
UsePreconditioner(true)

local i, j = W(), H()
local inside = eq(Mask(i, j), 0)
Offset:Exclude(Not(inside))
Angle:Exclude(Not(inside))

-- rigidity between a pixel and one neighbour, measured in the pixel's own rotated frame
local function rigidity(di, dj)
    local d_now  = Offset(i, j) - Offset(i + di, j + dj)
    local d_rest = UrShape(i, j) - UrShape(i + di, j + dj)
    local ok = InBounds(i + di, j + dj) * inside * eq(Mask(i + di, j + dj), 0)
    return Select(ok, w_regSqrt * (d_now - Rotate2D(Angle(i, j), d_rest)), 0)
end

local pinned = All(greatereq(Constraints(i, j), 0)) * inside
r = Residuals {
    reg_px = rigidity( 1,  0),
    reg_nx = rigidity(-1,  0),
    reg_py = rigidity( 0,  1),
    reg_ny = rigidity( 0, -1),
    fit    = w_fitSqrt * Select(pinned, Offset(i, j) - Constraints(i, j), 0.0)
}
