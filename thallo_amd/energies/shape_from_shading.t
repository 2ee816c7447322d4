-- Shape from shading: refine a depth map X so that the spherical-harmonics shading of its normals matches the
-- intensity image, while staying close to the measured depth D_i and locally smooth.
-- Same energy as the reference's examples/shape_from_shading/shape_from_shading.t; written for this repo.
local DISCONTINUITY = 0.01
local W, H = Dims("W", "H")
Inputs {
    w_p = Param(float, 0),  w_s = Param(float, 1),  w_g = Param(float, 2),      -- squared weights: fit, smoothness, shading
    f_x = Param(float, 3),  f_y = Param(float, 4),  u_x = Param(float, 5),  u_y = Param(float, 6),
    L_1 = Param(float, 7),  L_2 = Param(float, 8),  L_3 = Param(float, 9),  L_4 = Param(float, 10),  L_5 = Param(float, 11),
    L_6 = Param(float, 12), L_7 = Param(float, 13), L_8 = Param(float, 14), L_9 = Param(float, 15),
    X         = Unknown(thallo_float, {W, H}, 16),
    D_i       = Array(thallo_float, {W, H}, 17),
    Im        = Array(thallo_float, {W, H}, 18),
    edgeMaskR = Array(uint8, {W, H}, 19),
    edgeMaskC = Array(uint8, {W, H}, 20),
}
w_p, w_s, w_g = sqrt(w_p), sqrt(w_s), sqrt(w_g)
local x, y = W(), H()
local px, py = x:asvalue(), y:asvalue()

local function backproject(dx, dy)
    local d = X(x + dx, y + dy)
    return Vector(((dx + px - u_x) / f_x) * d, ((dy + py - u_y) / f_y) * d, d)
end
local function normal()
    local n_x = X(x, y - 1) * (X(x, y) - X(x - 1, y)) / f_y
    local n_y = X(x - 1, y) * (X(x, y) - X(x, y - 1)) / f_x
    local n_z = (n_x * (u_x - px) / f_x) + (n_y * (u_y - py) / f_y) - (X(x - 1, y) * X(x, y - 1) / (f_x * f_y))
    local len2 = n_x * n_x + n_y * n_y + n_z * n_z
    return Select(greater(len2, 0.0), 1.0 / sqrt(len2), 1.0) * Vector(n_x, n_y, n_z)
end
local function shading()
    local n = normal()
    local a, b, c = n[0], n[1], n[2]
    return L_1 + L_2 * b + L_3 * c + L_4 * a + L_5 * a * b + L_6 * b * c + L_7 * (-a * a - b * b + 2 * c * c) + L_8 * c * a + L_9 * (a * a - b * b)
end
local function depthOK(dx, dy) return greater(D_i(x + dx, y + dy), 0) end
local target = Im(x, y) * 0.5 + 0.25 * (Im(x - 1, y) + Im(x, y - 1))
local B_I_here = Select(depthOK(-1, 0) * depthOK(0, 0) * depthOK(0, -1), shading() - target, 0)
local function B_I(dx, dy) return B_I_here:get(x + dx, y + dy) end

local function near(dx, dy) return less(abs(X(x, y) - X(x + dx, y + dy)), DISCONTINUITY) end
local smooth = depthOK(0, 0) * depthOK(0, -1) * depthOK(0, 1) * depthOK(-1, 0) * depthOK(1, 0) * near(0, -1) * near(0, 1) * near(-1, 0) * near(1, 0)
smooth = eq(smooth:get(x, y), 1)
local laplacian = 4.0 * backproject(0, 0) - (backproject(-1, 0) + backproject(0, -1) + backproject(1, 0) + backproject(0, 1))

r = Residuals {
    fit       = Select(depthOK(0, 0), w_p * (X(x, y) - D_i(x, y)), 0),
    shading_h = Select(InBoundsExpanded(x, y, 1), w_g * ((B_I(0, 0) - B_I(1, 0)) * edgeMaskR(x, y)), 0),
    shading_v = Select(InBoundsExpanded(x, y, 1), w_g * ((B_I(0, 0) - B_I(0, 1)) * edgeMaskC(x, y)), 0),
    reg       = Select(smooth, w_s * laplacian, 0)
}
