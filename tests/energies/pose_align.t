-- Frame-to-frame alignment of corresponding points with one SE(3) pose per frame (the sparse term of the reference's sparse_bundle_fusion example):
-- PoseToMatrix (rotation vector + translation -> 4x4, Taylor forms near zero rotation) and rigid_trans from lib.t, both ends of a correspondence
-- reached through Sparse maps.
T, K = Dims("T", "K")
Inputs {
    Trans = Unknown(float3, {T}, 0),
    Rot   = Unknown(float3, {T}, 1),
    Pj    = Array(float3, {K}, 2),
    Pi    = Array(float3, {K}, 3),
    fi    = Sparse({K}, {T}, 4),
    fj    = Sparse({K}, {T}, 5)
}
UsePreconditioner(true)
local k = K()
local function pose(f) return PoseToMatrix(Rot(f), Trans(f)) end
r = Residuals {
    align = rigid_trans(pose(fi(k)), Pi(k)) - rigid_trans(pose(fj(k)), Pj(k))
}
