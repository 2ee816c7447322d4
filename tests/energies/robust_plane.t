-- Point-to-plane alignment with a per-vertex robustness weight that is itself an unknown (the structure of the reference's robust_nonrigid_alignment example):
-- two unknown arrays with different channel counts in one problem, a product of unknowns, a neighbour term through a Sparse map.  Written for this repo's tests.
local N = Dims("N")
Inputs {
    X   = Unknown(thallo_float3, {N}, 0),
    R   = Unknown(thallo_float, {N}, 1),
    T   = Array(thallo_float3, {N}, 2),
    Nrm = Array(thallo_float3, {N}, 3),
    w_rob = Param(float, 4),
    nb  = Sparse({N}, {N}, 5)
}
UsePreconditioner(true)
local i = N()
local d = dot(Nrm(i), X(i) - T(i))
Residuals {
    plane  = R(i) * R(i) * d,
    prior  = w_rob * (1.0 - R(i) * R(i)),
    smooth = 0.3 * (X(i) - X(nb(i)))
}
