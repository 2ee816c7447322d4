-- A linear fit written with Sum (lib.t:146): T(n) ~ sum_m S(n, m) W(m).  The construct of the reference's tests/minimal_fitting (own text, other names);
-- J p materialized like there.  The unknown W lives over M alone, the residual over N: every residual instance reads ALL of W.
local N, M = Dims("N", "M")
Inputs {
    Weights  = Unknown(float, {M}, 0),
    Basis    = Array(float, {N, M}, 1),
    Target   = Array(float, {N}, 2)
}
local n, m = N(), M()
local model = Sum({m}, Basis(n, m) * Weights(m))
local r = Residuals { fit = Target(n) - model }
r.fit.Jp:set_materialize(true)
