-- Two shared parameters (a, b) of y = a cos(b x) + b sin(a x) fitted to N samples; every residual reaches the ONE parameter pair through a Sparse map
-- (the scenario of the reference's tests/dense: all residuals scatter into the same two unknowns).  Written for this repo's tests.
local N, U, E = Dims("N", "U", "E")
Inputs {
    params  = Unknown(thallo_float2, {U}, 0),
    samples = Array(thallo_float2, {N}, 1),
    S       = Sparse({E}, {N}, 2),
    P       = Sparse({E}, {U}, 3)
}
UsePreconditioner(true)
local e = E()
local x, y = samples(S(e))(0), samples(S(e))(1)
local a, b = params(P(e))(0), params(P(e))(1)
Residuals { fit = y - (a * cos(b * x) + b * sin(a * x)) }
