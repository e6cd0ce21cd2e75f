#!/usr/bin/env python3
"""Generate the committed golden vectors of the hot path.

The reference is a Rust workspace and cannot be executed in the authoring image (no cargo/rustc),
and its own tests hold no stored matrix entries for this path. These vectors are
therefore produced by the CPU oracle (oracle/*.c, the line-by-line restatement of the reference) and
pin it against regressions; the oracle itself is pinned by the reference's known answers in
tests/test_oracle.py. Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

RADIUS, C = 0.1, 343.0


def k_from_ka(ka):
    k = ka / RADIUS
    return O.wave_number(k * C / (2.0 * np.pi), C)


def mixed_cube_sphere(radius, m):
    """6 m^2 quads of a cube's face grids projected onto the sphere; every third quad split into two triangles."""
    idx, nodes, conn = {}, [], []
    g = np.linspace(-1.0, 1.0, m + 1)

    def nid(p):
        key = tuple(np.round(p, 12))
        if key not in idx:
            idx[key] = len(nodes); nodes.append(p)
        return idx[key]
    for axis in range(3):
        for sign in (-1.0, 1.0):
            for a in range(m):
                for b in range(m):
                    quad = []
                    for (u, v) in ((g[a], g[b]), (g[a + 1], g[b]), (g[a + 1], g[b + 1]), (g[a], g[b + 1])):
                        p = np.zeros(3); p[axis] = sign; p[(axis + 1) % 3] = u; p[(axis + 2) % 3] = v
                        quad.append(nid(radius * p / np.linalg.norm(p)))
                    if sign < 0:
                        quad = quad[::-1]
                    if len(conn) % 3 == 0:
                        conn.append([quad[0], quad[1], quad[2], -1]); conn.append([quad[0], quad[2], quad[3], -1])
                    else:
                        conn.append(quad)
    return np.array(nodes), np.array(conn, dtype=np.int32)


def main():
    out = {}
    # --- 80-panel icosphere system at ka = 1 (sign -1, beta = 4i/k) and ka = 0.2 (sign +1, beta = i/k)
    om = O.icosphere(RADIUS, 1)
    out["ico1_nodes"] = om.nodes; out["ico1_conn"] = om.conn
    for tag, ka in (("ka1", 1.0), ("ka02", 0.2)):
        k = k_from_ka(ka)
        beta, _ = O.beta_adaptive(k, RADIUS)
        A, rhs0 = O.build_tbem_system_with_beta(om, k, beta)
        rhs = rhs0 + O.compute_rhs_with_beta(om.center, om.normal, k, beta)
        x, ipiv, rc = O.zgesv(A, rhs)
        assert rc == 0
        out["ico1_%s_k" % tag] = np.array([k]); out["ico1_%s_beta" % tag] = np.array([beta])
        out["ico1_%s_A" % tag] = A; out["ico1_%s_rhs" % tag] = rhs; out["ico1_%s_x" % tag] = x
    # --- raw panel integrals: far, near (subdivided) and self pairs on the 320-panel icosphere, ka = 1 and 3
    om2 = O.icosphere(RADIUS, 2)
    pairs = [(0, 200), (5, 17), (0, 1), (0, 2), (10, 11), (100, 101), (37, 37), (0, 0), (319, 319)]
    out["ico2_pairs"] = np.array(pairs, dtype=np.int32)
    for tag, ka in (("ka1", 1.0), ("ka3", 3.0)):
        k = k_from_ka(ka)
        res = []; nsub = []
        for i, j in pairs:
            if i == j:
                res.append(O.singular_integration(om2.center[i], om2.normal[i], om2.coords(i), k)); nsub.append(0)
            else:
                res.append(O.regular_integration(om2.center[i], om2.normal[i], om2.coords(j), om2.area[j], k))
                nsub.append(len(O.generate_subelements(om2.center[i], om2.coords(j), om2.area[j])))
        out["ico2_%s_integrals" % tag] = np.array(res); out["ico2_%s_nsub" % tag] = np.array(nsub, dtype=np.int32)
        out["ico2_%s_k" % tag] = np.array([k])
    # --- Mie series (rigid sphere, 50 terms) on the surface and at r = 2a
    theta = np.linspace(0.0, np.pi, 19)
    for tag, ka in (("ka02", 0.2), ("ka1", 1.0), ("ka3", 3.0)):
        k = k_from_ka(ka)
        out["mie_%s" % tag] = O.sphere_scattering_3d(k, RADIUS, 50, [RADIUS, 2 * RADIUS], theta)
    out["mie_theta"] = theta
    # --- UV sphere S10 geometry checksums (10 000 panels): enough to pin the generator without 240 KB of nodes
    s10 = O.uv_sphere(RADIUS, 51, 100)
    out["s10_checksums"] = np.array([s10.nodes.sum(), np.abs(s10.nodes).sum(), s10.area.sum(), s10.center[:, 2].sum(),
                                     float(s10.conn[:, :3].astype(np.int64).sum()), float(s10.n_elem)])
    # --- a mixed Tri3 + Quad4 mesh (cube-sphere, 2 x 2 faces, every third quad split) with nodal boundary values:
    #     matrix and boundary-value right-hand side at ka = 1, beta = 4i/k
    nodes, conn = mixed_cube_sphere(RADIUS, 2)
    mq = O.Mesh(nodes, conn)
    nn = np.where(mq.conn[:, 3] >= 0, 4, 3)
    mq.bc_len[:] = nn
    t = np.arange(mq.n_elem)[:, None] * 4 + np.arange(4)[None, :]
    mq.bc_values[:] = np.where(np.arange(4)[None, :] < nn[:, None], 1e-3 * (np.cos(0.7 * t) + 1j * np.sin(0.3 * t)), 0.0)
    mq.bc_type[5:9] = 1
    k = k_from_ka(1.0); beta = complex(0.0, 4.0 / k)
    Aq, rq = O.build_tbem_system_with_beta(mq, k, beta)
    out["mixq_nodes"] = mq.nodes; out["mixq_conn"] = mq.conn; out["mixq_bc_type"] = mq.bc_type; out["mixq_bc_len"] = mq.bc_len
    out["mixq_bc_values"] = mq.bc_values; out["mixq_k"] = np.array([k]); out["mixq_beta"] = np.array([beta]); out["mixq_A"] = Aq; out["mixq_rhs"] = rq
    np.savez_compressed(os.path.join(HERE, "bem_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "bem_golden.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
