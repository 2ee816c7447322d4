-- One rigid motion (Euler angles + translation) for a whole point set (the structure of the reference's procrustes_alignment example): six unknowns shared by every
-- residual, the residual over the product domain {N, U} with U = 1, points without a target skipped.  J and J^T J materialized: with six unknowns the dense
-- [JtJ]p schedule (J^T J formed on the matrix cores) runs in single precision.  Written for this repo's tests.
local N, U = Dims("N", "U")
Inputs {
    Shift = Unknown(thallo_float3, {U}, 0),
    Euler = Unknown(thallo_float3, {U}, 1),
    Rest  = Array(thallo_float3, {N}, 2),
    Goal  = Array(thallo_float3, {N}, 3)
}
UsePreconditioner(true)
local n, u = N(), U()
local has_goal = greatereq(Goal(n)(0), -999999.9)
r = Residuals { fit = Select(has_goal, Rotate3D(Euler(u), Rest(n)) + Shift(u) - Goal(n), 0) }
r.fit.J:set_materialize(true)
r.fit.JtJ:set_materialize(true)
