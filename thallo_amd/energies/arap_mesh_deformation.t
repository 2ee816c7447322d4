-- As-rigid-as-possible mesh deformation over directed edges (V0[e] -> V1[e]).
-- Same energy as the reference's examples/arap_mesh_deformation/arap_mesh_deformation.t.
N, E = Dims("N", "E")
Inputs {
    w_fitSqrt   = Param(float, 0),
    w_regSqrt   = Param(float, 1),
    Position    = Unknown(thallo_float3, {N}, 2),
    Angle       = Unknown(thallo_float3, {N}, 3),
    Original    = Array(thallo_float3, {N}, 4),
    Constraints = Array(thallo_float3, {N}, 5),   -- x < -999999.9 marks an unconstrained vertex
    V0          = Sparse({E}, {N}, 6),
    V1          = Sparse({E}, {N}, 7)
}
UsePreconditioner(true)
n, e = N(), E()
local a, b = V0(e), V1(e)
local handle = greatereq(Constraints(n)(0), -999999.9)
local edge_now  = Position(a) - Position(b)
local edge_rest = Original(a) - Original(b)
r = Residuals {
    fit = Select(handle, w_fitSqrt * (Position(n) - Constraints(n)), 0),
    reg = w_regSqrt * (edge_now - Rotate3D(Angle(a), edge_rest))
}
