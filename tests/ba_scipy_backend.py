"""CPU mirror of the SHARD form of the multi-GPU Gauss-Newton step (thallo_amd/csrc/solver_dist.cpp Plan::dist_gn_shard) for camera-sharded bundle
adjustment -- TEST INFRASTRUCTURE (oracle CSR of the rank's sub-instance + scipy, float64 accumulate / float32 state, torch.distributed / gloo).
The executable statement of that schedule: a rank holds its cameras, ALL points and the observations of its cameras; J^T F, diag(J^T J) and A p are
partial sums on the point block and are all-reduced; the vector update runs on [own cameras | all points], the points redundantly; the scalars are
the rank-ordered sum of the ranks' camera parts (one tiny all-gather) plus the point part every rank computes for itself after the all-reduce."""
import numpy as np
import scipy.sparse as sp
import torch
import torch.distributed as dist

from oracle import oracle as orc

F = np.float32


def _allreduce(vec, world):
    if world > 1:
        t = torch.from_numpy(vec)
        dist.all_reduce(t)
    return vec


def _allgather(vec, world):
    t = torch.from_numpy(np.ascontiguousarray(vec))
    out = [torch.empty_like(t) for _ in range(world)]
    if world > 1:
        dist.all_gather(out, t)
    else:
        out = [t]
    return [o.numpy() for o in out]


class BaShardMirror:
    def __init__(self, layout, local_params):
        self.lay = layout
        self.params = [a.copy() for a in local_params]
        cams, pts, obs, oc, op = self.params
        self.Cp, self.P, self.O = cams.shape[0], pts.shape[0], obs.shape[0]
        self.nc, self.n = 9 * self.Cp, 9 * self.Cp + 3 * self.P

    def _problem(self):
        return orc.Problem(orc.BUNDLE_ADJUST, (self.Cp, self.P, self.O), self.params)

    def cost(self):
        mine = np.array([self._problem().cost() if self.O else 0.0], np.float64)
        return float(sum(F(g[0]) for g in _allgather(mine, self.lay.world)))

    def _scalars(self, cam_parts, pt_parts):
        """rank-ordered sum of the camera parts + the point parts (identical on every rank)"""
        got = _allgather(np.asarray(cam_parts, np.float64), self.lay.world)
        return [sum(g[i] for g in got) + pt_parts[i] for i in range(len(cam_parts))]

    def gn_step(self, L):
        w, nc, n = self.lay.world, self.nc, self.n
        rp, col, val, res = self._problem().csr()
        J = sp.csr_matrix((val.astype(np.float64), col, rp), shape=(len(res), n))
        r = (-(J.T @ res.astype(np.float64))).astype(F)
        d = np.asarray(J.multiply(J).sum(0)).ravel().astype(F)
        _allreduce(r[nc:], w); _allreduce(d[nc:], w)                  # point blocks of J^T F and of the raw diagonal
        pre = (F(1) / (F(1) + np.sqrt(d)) ** 2).astype(F)
        z = pre * r
        dot = lambda a, b, sl: float(a[sl].astype(np.float64) @ b[sl].astype(np.float64))
        cam, pts = slice(0, nc), slice(nc, n)
        aN = F(F(self._scalars([dot(r, z, cam)], [0.0])[0]) + F(dot(r, z, pts)))
        p = np.zeros(n, F); delta = np.zeros(n, F); Ap = np.zeros(n, F)
        alpha = beta = F(0)
        for k in range(L):
            if k:
                r = (r - alpha * Ap).astype(F); delta = (delta + alpha * p).astype(F)
            p = (pre * r + beta * p).astype(F)
            Ap = (J.T @ (J @ p.astype(np.float64))).astype(F)
            _allreduce(Ap[nc:], w)                                    # the point block of A p: the per-iteration all-reduce
            m64, r64, a64, p64 = (v.astype(np.float64) for v in (pre, r, Ap, p))
            sums = lambda sl: [float(p64[sl] @ a64[sl]), float((m64[sl] * r64[sl]) @ r64[sl]), float((m64[sl] * r64[sl]) @ a64[sl]), float((m64[sl] * a64[sl]) @ a64[sl])]
            c, q = sums(cam), sums(pts)
            tot = self._scalars(c, [0.0] * 4)                         # ONE tiny all-gather: the camera parts, rank order
            aD = F(F(tot[0]) + F(q[0])); n_, s1, s2 = tot[1] + q[1], tot[2] + q[2], tot[3] + q[3]
            alpha = aN / aD if aD != 0 else F(0)
            bN = F(max(n_ - 2.0 * float(alpha) * s1 + float(alpha) ** 2 * s2, 0.0))
            beta = bN / aN if aN != 0 else F(0)
            aN = bN
        if L:
            delta = (delta + alpha * p).astype(F)
        self.params[0].reshape(-1)[:] += delta[:nc]
        self.params[1].reshape(-1)[:] += delta[nc:]

    def solve(self, nit, lit):
        costs = [self.cost()]
        for _ in range(nit):
            self.gn_step(lit)
            costs.append(self.cost())
        return costs

    # ---- Levenberg-Marquardt on the same shards (round 6; solver_dist.cpp Plan::step_lm_shard): every element-wise LM kernel on the camera block and the point block
    # separately -- the camera sums travel in one tiny all-gather (rank order), the point sums are added by every rank for itself --, (J^T J) p and (J^T J) delta
    # all-reduced on the point block BEFORE CtC p enters, the two sums of the model cost linear in the ranks' contributions, accept / revert and the trust region
    # replicated on identical scalars (gauss_newton.t:1545-1785 with every UsesLambda() branch taken; oracle/thallo_oracle.c orc_solve)
    def lm_solve(self, nit, lit, **kw):
        sp = orc.default_params(**kw)
        w, nc, n = self.lay.world, self.nc, self.n
        cam, pts = slice(0, nc), slice(nc, n)
        dot = lambda a, b, sl: float(a[sl].astype(np.float64) @ b[sl].astype(np.float64))
        gsum = lambda pairs: [F(F(t) + F(q)) for t, q in zip(self._scalars([c for c, _ in pairs], [0.0] * len(pairs)), [q for _, q in pairs])]      # float words, as the device forms them
        radius, dec = F(sp.trust_region_radius), F(sp.radius_decrease_factor)
        prev = F(self.cost()); costs = [float(prev)]
        SSq = None
        for it in range(nit):
            rp, col, val, res = self._problem().csr()
            J = sp_csr(val, col, rp, len(res), n)
            r = (-(J.T @ res.astype(np.float64))).astype(F)
            d = np.asarray(J.multiply(J).sum(0)).ravel().astype(F)
            _allreduce(r[nc:], w); _allreduce(d[nc:], w)
            if it == 0:
                SSq = (F(1) / (F(1) + np.sqrt(d)) ** 2).astype(F)              # PCGSaveSSq: guardedInvert of the raw diagonal (use_preconditioner)
            unclamped = (d * (F(1) / radius)).astype(F)
            cm = ((F(1) / SSq) / radius).astype(F)
            CtC = np.minimum(np.maximum(unclamped, F(sp.min_lm_diagonal) * cm), F(sp.max_lm_diagonal) * cm).astype(F)
            pre = (F(1) / (CtC + radius * unclamped)).astype(F)
            b = r.copy(); z = (pre * r).astype(F)
            aN = gsum([(dot(r, z, cam), dot(r, z, pts))])[0]
            p = np.zeros(n, F); delta = np.zeros(n, F)
            Q0 = F(0); beta = F(0)
            for k in range(sp.lIterations if False else lit):
                p = (z + beta * p).astype(F) if k else z.copy()
                Ap = (J.T @ (J @ p.astype(np.float64))).astype(F)
                _allreduce(Ap[nc:], w)
                Ap = (Ap + CtC * p).astype(F)                                  # PCGStep1_Finish behind the all-reduce: CtC p enters once
                aD = gsum([(dot(p, Ap, cam), dot(p, Ap, pts))])[0]
                alpha = F(aN / aD)
                delta = (delta + alpha * p).astype(F)
                if (k + 1) % sp.residual_reset_period == 0:                    # :1653-1657
                    Ad = (J.T @ (J @ delta.astype(np.float64))).astype(F)
                    _allreduce(Ad[nc:], w)
                    Ad = (Ad + delta * CtC).astype(F)
                    r = (b - Ad).astype(F)
                else:
                    r = (r - alpha * Ap).astype(F)
                z = (pre * r).astype(F)
                rb = (r + b).astype(F)
                bN, Q1 = gsum([(dot(z, r, cam), dot(z, r, pts)), (0.5 * dot(delta, rb, cam), 0.5 * dot(delta, rb, pts))])
                beta = F(bN / aN); aN = bN
                if not np.isfinite(Q1): break
                zeta = F(k + 1) * (Q1 - Q0) / Q1
                if not np.isfinite(zeta) or zeta < F(sp.q_tolerance): break
                Q0 = Q1
            Ad = (J.T @ (J @ delta.astype(np.float64)))
            # the two sums of the model cost: linear in the ranks' contributions (delta . this rank's part of J^T J delta added over the ranks; delta . b: cameras + points once)
            dJJd, db = gsum([(float(delta.astype(np.float64) @ Ad), 0.0), (dot(delta, b, cam), dot(delta, b, pts))])
            prevX = [self.params[0].copy(), self.params[1].copy()]
            self.params[0].reshape(-1)[:] += delta[:nc]
            self.params[1].reshape(-1)[:] += delta[nc:]
            new = F(self.cost())
            model = F(db - F(0.5) * dJJd)
            change = F(prev - new); rel = F(change / model)
            if change >= 0 and rel > F(sp.min_relative_decrease):
                if change <= prev * F(sp.function_tolerance):
                    costs.append(float(new)); break
                tmp = 1.0 - (2.0 * float(rel) - 1.0) ** 3
                radius = F(min(float(F(float(radius) / max(1.0 / 3.0, tmp))), sp.max_trust_region_radius)); dec = F(2); prev = new
            else:
                self.params[0][:] = prevX[0]; self.params[1][:] = prevX[1]
                radius = F(radius / dec); dec = F(2 * dec)
                if radius < F(sp.min_trust_region_radius):
                    costs.append(float(prev)); break
            costs.append(float(F(self.cost())))
        return costs


def sp_csr(val, col, rp, rows, n):
    return sp.csr_matrix((val.astype(np.float64), col, rp), shape=(rows, n))
