-- Embedded deformation (the structure of the reference's embedded_mesh_deformation example): every node carries an offset and a free 3 x 3 matrix (nine unknown
-- channels), neighbouring nodes must agree on where the edge between them goes, and a six-component residual pulls each matrix towards a rotation.
-- Written for this repo's tests.
local N, E = Dims("N", "E")
Inputs {
    w_fit = Param(float, 0),
    w_reg = Param(float, 1),
    w_rot = Param(float, 2),
    Off   = Unknown(thallo_float3, {N}, 3),
    M     = Unknown(thallo_float9, {N}, 4),
    Rest  = Array(thallo_float3, {N}, 5),
    Goal  = Array(thallo_float3, {N}, 6),
    a     = Sparse({E}, {N}, 7),
    b     = Sparse({E}, {N}, 8)
}
UsePreconditioner(true)
local n, e = N(), E()
local pinned = greatereq(Goal(n)(0), -999999.9)
local edge = (Off(b(e)) - Off(a(e))) - gemv(M(a(e)), Rest(b(e)) - Rest(a(e)))
local R = M(n)
local c0, c1, c2 = Vector(R(0), R(3), R(6)), Vector(R(1), R(4), R(7)), Vector(R(2), R(5), R(8))
Residuals {
    fit = Select(pinned, w_fit * (Off(n) - Goal(n)), 0),
    reg = w_reg * edge,
    rot = { w_rot * dot(c0, c1), w_rot * dot(c0, c2), w_rot * dot(c1, c2), w_rot * (dot(c0, c0) - 1), w_rot * (dot(c1, c1) - 1), w_rot * (dot(c2, c2) - 1) }
}
