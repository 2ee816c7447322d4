-- As-rigid-as-possible deformation of a regular 3-D lattice (the structure of the reference's volumetric_mesh_deformation example): a position and an
-- Euler-angle triple per lattice node, six neighbours, some nodes tied to targets.  A three-dimensional iteration domain.  Written for this repo's tests.
local W, H, D = Dims("W", "H", "D")
Inputs {
    w_fit = Param(float, 0),
    w_reg = Param(float, 1),
    Pos  = Unknown(thallo_float3, {W, H, D}, 2),
    Ang  = Unknown(thallo_float3, {W, H, D}, 3),
    Rest = Array(thallo_float3, {W, H, D}, 4),
    Tgt  = Array(thallo_float3, {W, H, D}, 5)
}
UsePreconditioner(true)
local x, y, z = W(), H(), D()
local function edge(dx, dy, dz)
    local e = (Pos(x, y, z) - Pos(x + dx, y + dy, z + dz)) - Rotate3D(Ang(x, y, z), Rest(x, y, z) - Rest(x + dx, y + dy, z + dz))
    return Select(InBounds(x + dx, y + dy, z + dz), w_reg * e, 0)
end
Residuals {
    fit = Select(greatereq(Tgt(x, y, z)(0), -999999.9), w_fit * (Pos(x, y, z) - Tgt(x, y, z)), 0),
    e_px = edge(1, 0, 0), e_mx = edge(-1, 0, 0),
    e_py = edge(0, 1, 0), e_my = edge(0, -1, 0),
    e_pz = edge(0, 0, 1), e_mz = edge(0, 0, -1)
}
