-- The same fit as curve_fit_graph.t with the residual over the product domain {N, U} (U = 1): the unknown is indexed by its own dimension, not through a map.
local N, U = Dims("N", "U")
Inputs {
    params  = Unknown(thallo_float2, {U}, 0),
    samples = Array(thallo_float2, {N}, 1)
}
UsePreconditioner(true)
local n, u = N(), U()
local x, y = samples(n)(0), samples(n)(1)
local a, b = params(u)(0), params(u)(1)
Residuals { fit = y - (a * cos(b * x) + b * sin(a * x)) }
