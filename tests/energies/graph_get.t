-- A per-vertex expression read through the graph's Sparse maps with :get (tests/minimal_sparse_materialize in the reference): sin(X) at both ends
-- of every edge.
N, E = Dims("N", "E")
Inputs {
    X = Unknown(float, {N}, 0),
    A = Array(float, {N}, 1),
    v0 = Sparse({E}, {N}, 2),
    v1 = Sparse({E}, {N}, 3)
}
n, e = N(), E()
local wave = sin(X(n))
r = Residuals {
    fit = X(n) - A(n),
    reg = wave:get(v0(e)) - wave:get(v1(e))
}
