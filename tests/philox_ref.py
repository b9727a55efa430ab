"""numpy restatement of the device noise generator (csrc/mdx_transition.hip: philox_noise_kernel).

Philox4x32-10 (Salmon et al., SC'11), key = 64-bit seed, counter =
(local element index [*2 + block for the categorical draws], draw index, molecule id low, (molecule id high << 4) | stream),
stream 0 = eps_pos, 1 = u_node, 2 = u_halfedge.  Uniform = (x >> 8) * 2^-24; normals by Box-Muller on
((x>>8)+1)*2^-24 and (y>>8)*2^-24.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c, k0, k1):
    """c: (n,4) uint32 counters -> (n,4) uint32."""
    c = c.astype(np.uint64)
    k0, k1 = int(k0), int(k1)
    for _ in range(10):
        p0 = M0 * c[:, 0]
        p1 = M1 * c[:, 2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c = np.stack([hi1 ^ c[:, 1] ^ np.uint64(k0), lo1, hi0 ^ c[:, 3] ^ np.uint64(k1), lo0], 1)
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c.astype(np.uint32)


def u01(x):
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def noise_ref(seed, draw, sizes, mol_ids, Kn, Ke):
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    node_mol, node_loc, he_mol, he_loc = [], [], [], []
    for n, mid in zip(sizes, mol_ids):
        n = max(int(n), 0)
        node_mol += [int(mid)] * n
        node_loc += list(range(n))
        nh = n * (n - 1) // 2
        he_mol += [int(mid)] * nh
        he_loc += list(range(nh))

    def ctr(loc, mol, stream):
        mol = np.asarray(mol, dtype=np.uint64)
        lo = (mol & MASK).astype(np.uint32)
        hi = (((mol >> np.uint64(32)) << np.uint64(4)) & MASK).astype(np.uint32) | np.uint32(stream)
        return np.stack([np.asarray(loc, dtype=np.uint32), np.full(len(loc), draw, dtype=np.uint32), lo, hi], 1)

    N, Eh = len(node_loc), len(he_loc)
    eps = np.zeros((N, 3), np.float32)
    if N:
        r = philox4x32_10(ctr(node_loc, node_mol, 0), k0, k1)
        u1 = ((r[:, 0] >> np.uint32(8)).astype(np.float32) + np.float32(1)) * np.float32(2.0 ** -24)
        u3 = ((r[:, 2] >> np.uint32(8)).astype(np.float32) + np.float32(1)) * np.float32(2.0 ** -24)
        r1 = np.sqrt(np.float32(-2) * np.log(u1.astype(np.float64))).astype(np.float32)
        r2 = np.sqrt(np.float32(-2) * np.log(u3.astype(np.float64))).astype(np.float32)
        a1 = np.float32(6.283185307179586) * u01(r[:, 1])
        a2 = np.float32(6.283185307179586) * u01(r[:, 3])
        eps[:, 0] = r1 * np.cos(a1.astype(np.float64))
        eps[:, 1] = r1 * np.sin(a1.astype(np.float64))
        eps[:, 2] = r2 * np.cos(a2.astype(np.float64))

    def cat(loc, mol, stream, K, n):
        out = np.zeros((n, K), np.float32)
        if n == 0:
            return out
        for b in range((K + 3) // 4):
            r = philox4x32_10(ctr(np.asarray(loc, dtype=np.uint32) * np.uint32(2) + np.uint32(b), mol, stream), k0, k1)
            w = min(4, K - 4 * b)
            out[:, 4 * b:4 * b + w] = u01(r[:, :w])
        return out

    return eps, cat(node_loc, node_mol, 1, Kn, N), cat(he_loc, he_mol, 2, Ke, Eh)
