-- test energy for SampledImage: a two-channel target sampled at (pixel + flow), against a one-channel source; forward-difference smoothness.
-- Uses both call forms of a sampled image: (x, y) -> vector of channels, (x, y, c) -> one channel.
local W, H = Dims("W", "H")
Inputs {
    w_fit = Param(float, 0),
    w_reg = Param(float, 1),
    Flow  = Unknown(thallo_float2, {W, H}, 2),
    Src   = Array(thallo_float, {W, H}, 3),
    Dst   = Array(thallo_float2, {W, H}, 4),
    DstDx = Array(thallo_float2, {W, H}, 5),
    DstDy = Array(thallo_float2, {W, H}, 6)
}
local D = SampledImage(Dst, DstDx, DstDy)
local x, y = W(), H()
local px = x:asvalue() + Flow(x, y)(0)
local py = y:asvalue() + Flow(x, y)(1)
local both = D(px, py)
Residuals {
    fit0 = w_fit * (Src(x, y) - both(0)),
    fit1 = w_fit * (0.5 * Src(x, y) - D(px, py, 1)),
    smooth_x = Select(InBounds(x + 1, y), w_reg * (Flow(x, y) - Flow(x + 1, y)), 0),
    smooth_y = Select(InBounds(x, y + 1), w_reg * (Flow(x, y) - Flow(x, y + 1)), 0)
}
