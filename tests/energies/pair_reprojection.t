-- Pairwise frame-to-frame reprojection with one SE(3) pose per frame: the constructs of the dense term of the reference's examples/bundle_fusion_solve on a problem small
-- enough to restate in numpy.  Two iteration variables over ONE dimension (t0, t1 = T(), T()) re-bound through two Sparse maps by a positional :get; a one-variable
-- :get that RENAMES the variable; maps declared over another dimension (C) than the one they are indexed with (N), which the reference accepts; InvertRigidTransform /
-- matmul / Mat4ToRigidTransform / rigid_trans on the 3 x 4 form; CameraToDepth; SelectOnAll with -inf; Max; Constant; and a Sparse map's entry used as the layer of a
-- SampledImageArray (:asvalue()).
local T, C, N, LW, LH = Dims("T", "C", "N", "LW", "LH")
Inputs {
    Trans = Unknown(float3, {T}, 0),
    Rot   = Unknown(float3, {T}, 1),
    Q     = Array(float3, {N}, 2),
    Obs   = Array(float2, {N}, 3),
    Flag  = Array(float, {N}, 4),
    Depth = Array(float, {LW, LH, T}, 5),
    fx = Param(float, 6), fy = Param(float, 7), cx = Param(float, 8), cy = Param(float, 9),
    dmin = Param(float, 10), dmax = Param(float, 11), w_depth = Param(float, 12),
    src = Sparse({C}, {T}, 13),
    tgt = Sparse({C}, {T}, 14)
}
UsePreconditioner(true)
local Layer = SampledImageArray(Depth)
local n = N()
local t_s, t_t = src(n), tgt(n)
local t0, t1 = T(), T()
local pose = function(t) return PoseToMatrix(Rot(t0), Trans(t0)):get(t) end
local function relative(i, j) return Mat4ToRigidTransform(matmul(InvertRigidTransform(pose(i)), pose(j))) end
local M = relative(t0, t1):get(t_t, t_s)               -- source frame -> target frame
local X = rigid_trans(M, Q(n))
local uv = CameraToDepth(fx, fy, cx, cy, X)
local ok = { greater(X(2), dmin), less(X(2), dmax), neq(Flag(n), -inf) }
local wgt = Sqrt(Max(0.0, 1.0 - Constant(X(2)) / 8.0))
local uvc = CameraToDepth(fx, fy, cx, cy, Constant(X))
local d = Layer(uvc(0), uvc(1), t_t:asvalue())
r = Residuals {
    reproj = SelectOnAll(ok, wgt * (uv - Obs(n)), 0.0),
    depth  = SelectOnAll(ok, w_depth * (d - X(2)), 0.0)
}
