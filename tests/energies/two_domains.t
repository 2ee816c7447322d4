-- Two unknown arrays over two different index spaces in one problem (the scenario of the reference's tests/multidomain): a handful of shifts S(u) and a value per
-- point P(n); the fit term lives on the product domain {N, U}, the prior on {N} alone.  Written for this repo's tests.
local N, U = Dims("N", "U")
Inputs {
    S = Unknown(thallo_float, {U}, 0),
    P = Unknown(thallo_float, {N}, 1),
    T = Array(thallo_float, {N}, 2)
}
UsePreconditioner(true)
local n, u = N(), U()
Residuals {
    fit  = S(u) + P(n) - T(n),
    keep = 0.5 * P(n)
}
