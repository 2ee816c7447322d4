-- test energy for SampledImageArray (lib.t:145): a stack of T images sampled bilinearly at a shifted pixel position in layer `layer`; the sample carries no
-- derivative (as in the reference, thallo.t:5913-5916), so the energy is linear in U and one Gauss-Newton step lands on the samples.
local W, H, T = Dims("W", "H", "T")
Inputs {
    U     = Unknown(float, {W, H}, 0),
    Stack = Array(thallo_float2, {W, H, T}, 1),
    layer = Param(float, 2),
    sx    = Param(float, 3),
    sy    = Param(float, 4)
}
local S = SampledImageArray(Stack)
local x, y = W(), H()
local both = S(x:asvalue() + sx, y:asvalue() + sy, layer)
Residuals {
    fit = U(x, y) - (both(0) + 2.0 * S(x:asvalue() + sx, y:asvalue() + sy, layer, 1))
}
