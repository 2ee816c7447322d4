-- Bundle adjustment with the Snavely/Bundler camera model (9 parameters per camera:
-- angle-axis rotation, translation, focal length, two radial distortion coefficients).
-- Same energy as the reference's examples/bundle_adjustment/bundle_adjustment.t; written for this repo.
local C, P, O = Dims("C", "P", "O")
Inputs {
    cameras      = Unknown(thallo_float9, {C}, 0),
    points       = Unknown(thallo_float3, {P}, 1),
    observations = Array(float2, {O}, 2),
    oToC         = Sparse({O}, {C}, 3),     -- observation -> camera
    oToP         = Sparse({O}, {P}, 4)      -- observation -> point
}
UsePreconditioner(true)
local o = O()
local cam, X = cameras(oToC(o)), points(oToP(o))
local inCamera = AngleAxisRotatePoint(cam:slice(0, 3), X) + cam:slice(3, 6)
-- Bundler's camera looks down the negative z axis
local centre = Vector(-inCamera(0) / inCamera(2), -inCamera(1) / inCamera(2))
local rr = dot(centre, centre)
local radial = 1.0 + rr * (cam(7) + cam(8) * rr)
local projected = centre * cam(6) * radial
r = Residuals { snavely_reprojection_error = observations(o) - projected }
