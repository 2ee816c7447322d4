-- Graph-domain Laplacian smoothing over an edge list (v0[e], v1[e]).
-- Same energy as the reference's tests/minimal_graph/laplacian.t.
local N, E = Dims("N", "E")
Inputs {
    X  = Unknown(float, {N}, 0),
    A  = Array(float, {N}, 1),
    v0 = Sparse({E}, {N}, 2),
    v1 = Sparse({E}, {N}, 3)
}
w_fit = 0.5
n, e = N(), E()
r = Residuals {
    fit = w_fit * (X(n) - A(n)),
    reg = X(v0(e)) - X(v1(e))
}
