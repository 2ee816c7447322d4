-- A 1-D deconvolution kernel fit: T(n) ~ sum_k Signal(n - k + 2) C(k), n away from the border -- index arithmetic between TWO iteration variables inside a Sum
-- (the construct of the reference's tests/convolution; own text).
local N, K = Dims("N", "K")
Inputs {
    Kernel = Unknown(float, {K}, 0),
    Signal = Array(float, {N}, 1),
    Target = Array(float, {N}, 2)
}
local n, k = N(), K()
local conv = Sum({k}, Signal(n - k + 2) * Kernel(k))
local e = Select(InBoundsExpanded(n, 2), Target(n) - conv, 0.0)
local r = Residuals { conv = e }
r.conv.Jp:set_materialize(true)
