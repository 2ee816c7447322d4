-- Mesh fairing with cotangent weights that depend on the unknown positions themselves (the structure of the reference's cotangent_mesh_smoothing example):
-- an interior edge (v0, v1) with the two vertices v2, v3 opposite to it pulls its end points together with the mean cotangent of the two opposite angles.
-- Four Sparse maps from the edge domain into the vertices; nonlinear through dot / cross / sqrt and a division.  Written for this repo's tests.
local N, E = Dims("N", "E")
Inputs {
    w_fit = Param(float, 0),
    w_reg = Param(float, 1),
    X  = Unknown(thallo_float3, {N}, 2),
    A  = Array(thallo_float3, {N}, 3),
    v0 = Sparse({E}, {N}, 4),
    v1 = Sparse({E}, {N}, 5),
    v2 = Sparse({E}, {N}, 6),
    v3 = Sparse({E}, {N}, 7)
}
UsePreconditioner(true)
local n, e = N(), E()
local function cot(p, q, apex)
    local a, b = p - apex, q - apex
    local c = cross(a, b)
    return dot(a, b) / sqrt(dot(c, c))
end
local p0, p1, p2, p3 = X(v0(e)), X(v1(e)), X(v2(e)), X(v3(e))
local wgt = 0.5 * (cot(p0, p1, p2) + cot(p0, p1, p3))
Residuals {
    fit = w_fit * (X(n) - A(n)),
    fair = w_reg * wgt * (p1 - p0)
}
