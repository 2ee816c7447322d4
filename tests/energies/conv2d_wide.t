-- Non-blind deconvolution with an 11 x 11 kernel (the structure of the reference's spatially_varying_deconvolution example at a size where a residual reads 121
-- unknowns): blur(x, y) = sum over the window of Ker(kx, ky) X(x + kx - 5, y + ky - 5) against the observed image, inside the border; a small prior keeps
-- the border pixels determined.  More unknown accesses per residual than forward-mode duals carry at once: the front-end's wide lowering.  Written for this repo's tests.
local W, H, KX, KY = Dims("W", "H", "KX", "KY")
Inputs {
    X   = Unknown(thallo_float, {W, H}, 0),
    B   = Array(thallo_float, {W, H}, 1),
    Ker = Array(thallo_float, {KX, KY}, 2)
}
UsePreconditioner(true)
local x, y, kx, ky = W(), H(), KX(), KY()
local blur = Sum({kx, ky}, Ker(kx, ky) * X(x + kx - 5, y + ky - 5))
Residuals {
    data  = Select(InBoundsExpanded(x, y, 5), blur - B(x, y), 0.0),
    prior = 0.05 * X(x, y)
}
