-- Gradient-domain pasting of a four-channel image (the kind of energy the reference's poisson_image_editing example states): inside the region the
-- differences of X to its four neighbours follow the differences of the guide image T; pixels outside the region (M != 0) keep their values and act
-- as boundary conditions.  Linear.  Written for this repo's tests.
local W, H = Dims("W", "H")
Inputs {
    X = Unknown(thallo_float4, {W, H}, 0),
    T = Array(thallo_float4, {W, H}, 1),
    M = Array(float, {W, H}, 2)
}
UsePreconditioner(false)
local x, y = W(), H()
X:Exclude(Not(eq(M(x, y), 0)))
local function grad(dx, dy)
    return Select(InBounds(x + dx, y + dy), (X(x, y) - X(x + dx, y + dy)) - (T(x, y) - T(x + dx, y + dy)), 0)
end
Residuals { e_right = grad(1, 0), e_left = grad(-1, 0), e_down = grad(0, 1), e_up = grad(0, -1) }
