"""Generates tests/golden/golden_v1.npz: known-answer vectors for the hot path from an
independent NumPy restatement of the reference's algorithm (not from the C oracle, and not
from the reference, which cannot be built or imported here: it is C++ needing fftw_cpp.hh,
fftw3 and glm).  The C oracle and the HIP path are both checked against these vectors.

Run from the repo root:  python tests/golden/make_golden.py
The output is committed; tests never regenerate it.

Citations are file:line in the reference tree.
"""
import os

import numpy as np

f32 = np.float32
W = 32768
SR = 44100
HERE = os.path.dirname(os.path.abspath(__file__))

PMAX = {0: [3520.0, 8.0, 3520.0, 1.0],
        1: [3520.0, 8.0, 3520.0, 8.0, 3520.0, 8.0],
        2: [3520.0, 8.0, 3520.0, 1.0],
        3: [3520.0, 8.0, 3520.0, 8.0, 3520.0, 8.0, 3520.0, 8.0]}
DIMS = {0: 4, 1: 6, 2: 12, 3: 8}


def wavetable():
    # Evolutionary_Strategy.hpp:328-331 with sin evaluated in double and rounded once
    i = np.arange(W, dtype=f32)
    inv = f32(1.0) / (f32(W) - f32(1.0))
    arg = ((i * inv) * f32(2)) * f32(np.pi)
    return np.sin(arg.astype(np.float64)).astype(f32)


def window(n):
    # Evolutionary_Strategy.hpp:308-317
    one_over = f32(1.0) / f32(n)
    w = 1.0 - np.cos(np.arange(n, dtype=np.float64) * np.float64(one_over - f32(1)) * (2.0 * np.pi))
    fac = f32(0)
    for x in w:
        fac = f32(np.float64(fac) + x)
    fac = f32(fac * one_over)
    return w, fac


def tab_at(tab, pos):
    i = int(pos)  # truncation, as (unsigned int)pos for in-range phases
    return tab[min(max(i, 0), W - 1)]


def wrap_hi(p):
    return f32(p - f32(W)) if p >= f32(W) else p


def wrap_lo(p):
    return f32(p + f32(W)) if p < f32(0) else p


def synth(kind, values, n, tab, pmin=None):
    pmax = PMAX[kind]
    d = DIMS[kind]
    pmin = [0.0] * len(pmax) if pmin is None else pmin
    c = f32(W) / f32(SR)  # w2srRatio, Evolutionary_Strategy.hpp:203
    p = []
    for g in range(d):
        s = g & 3 if kind == 2 else g
        p.append(f32(f32(pmin[s]) + f32(f32(values[g]) * f32(f32(pmax[s]) - f32(pmin[s])))))
    out = np.zeros(n, f32)
    if kind == 0:      # Evolutionary_Strategy.hpp:368-402
        mod, fc, amp = f32(p[0] * p[1]), p[2], p[3]
        inc = f32(c * p[0])
        p1 = p2 = f32(0)
        for i in range(n):
            cur = f32(f32(tab_at(tab, p1) * mod) + fc)
            p1 = wrap_hi(f32(p1 + inc))
            out[i] = f32(tab_at(tab, p2) * amp)
            p2 = wrap_lo(wrap_hi(f32(p2 + f32(c * cur))))
    elif kind in (1, 3):  # Evolutionary_Strategy.hpp:403-449 (+ one stage for the 4-op voice)
        ops = 3 if kind == 1 else 4
        m = [f32(p[2 * o] * p[2 * o + 1]) for o in range(ops)]
        inc = f32(c * p[1])
        pos = [f32(0)] * ops
        for i in range(n):
            cur = f32(f32(tab_at(tab, pos[0]) * m[0]) + p[3])
            pos[0] = wrap_hi(f32(pos[0] + inc))
            for o in range(1, ops - 1):
                nxt = f32(f32(tab_at(tab, pos[o]) * m[o]) + p[2 * o + 3])
                pos[o] = wrap_lo(wrap_hi(f32(pos[o] + f32(c * cur))))
                cur = nxt
            out[i] = f32(tab_at(tab, pos[ops - 1]) * m[ops - 1])
            pos[ops - 1] = wrap_lo(wrap_hi(f32(pos[ops - 1] + f32(c * cur))))
    else:              # Evolutionary_Strategy.hpp:450-495
        mod = [f32(p[4 * j] * p[4 * j + 1]) for j in range(3)]
        fc = [p[4 * j + 2] for j in range(3)]
        amp = [p[4 * j + 3] for j in range(3)]
        inc = [f32(c * p[4 * j]) for j in range(3)]
        pa = [f32(0)] * 3
        pb = [f32(0)] * 3
        for i in range(n):
            tot = []
            for j in range(3):
                cur = f32(f32(tab_at(tab, pa[j]) * mod[j]) + fc[j])
                pa[j] = wrap_hi(f32(pa[j] + inc[j]))
                tot.append(f32(tab_at(tab, pb[j]) * amp[j]))
                pb[j] = wrap_lo(wrap_hi(f32(pb[j] + f32(c * cur))))
            out[i] = f32(np.float64(f32(f32(tot[0] + tot[1]) + tot[2])) / 3.0)
    return out


def spectrum(audio, win, wf):
    # Evolutionary_Strategy.hpp:503-523
    n = len(audio)
    x = np.fft.rfft(audio.astype(np.float64) * win)
    raw = np.hypot(x.real.astype(f32), x.imag.astype(f32)).astype(f32)
    return ((raw * (f32(1.0) / f32(n))) * (f32(1.0) / wf))[: n // 2].astype(f32)


def fitness(mag, tgt):
    # Evolutionary_Strategy_CPU.hpp:230-265
    e = f32(0)
    for a, b in zip(mag, tgt):
        t = f32(a - b)
        e = f32(e + f32(t * t))
    return e


# ---- Philox4x32-10, vectorised over counters -------------------------------------------
def philox(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = [np.asarray(x, np.uint64) & 0xFFFFFFFF for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & np.uint64(0xFFFFFFFF)
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & np.uint64(0xFFFFFFFF)
        c0, c1, c2, c3 = n0, p1 & np.uint64(0xFFFFFFFF), n2, p0 & np.uint64(0xFFFFFFFF)
        k0 = (k0 + np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
        k1 = (k1 + np.uint64(0xBB67AE85)) & np.uint64(0xFFFFFFFF)
    return np.stack([c0, c1, c2, c3], -1).astype(np.uint32)


def draw(seed, gid, epoch, index, tag):
    r = philox(gid, epoch, index >> 2, tag, seed & 0xFFFFFFFF, seed >> 32)
    return np.take_along_axis(r, (np.asarray(index) & 3)[..., None].astype(np.int64), -1)[..., 0]


def unit(w):
    return w.astype(np.int32).astype(f32) / f32(2147483647.0)


TAG_INIT, TAG_MUT = 0x494E4954, 0x4D555441


def init_population(p, d, seed, gid_base, chunk):
    gid = (gid_base + np.arange(p))[:, None]
    j = np.arange(d)[None, :]
    u = unit(draw(seed, gid, chunk, np.broadcast_to(j, (p, d)), TAG_INIT))
    return np.abs(u).astype(f32), np.full((p, d), f32(0.1), f32)


def recombine(v, s, parents, block):
    # ocl_program.cl:99-148 with all reads before all writes
    p, d = v.shape
    npb = max(1, parents // block)
    vo, so = np.empty_like(v), np.empty_like(s)
    for b in range(p // block):
        pb = b % npb
        for l in range(block):
            for g in range(d):
                dst = b * block + (l + g * (b + 1)) % block
                vo[dst, g] = v[pb * block + l, g]
                so[dst, g] = s[pb * block + l, g]
    return vo, so


def mutate(v, s, seed, gid_base, gen):
    # ocl_program.cl:166-189, constants Evolutionary_Strategy.hpp:611-627
    p, d = v.shape
    alpha = f32(1.4)
    inv_alpha = f32(1.0) / alpha
    rtop = np.sqrt(f32(2.0) / f32(np.pi)).astype(f32)
    bscale = f32(1.0) / f32(d)
    beta = np.sqrt(bscale).astype(f32)
    gid = np.broadcast_to((gid_base + np.arange(p))[:, None], (p, d))
    base = np.broadcast_to((np.arange(d) * 16)[None, :], (p, d))
    w0 = draw(seed, gid, gen, base, TAG_MUT)
    ek = np.where(w0 % 2 == 0, alpha, inv_alpha).astype(f32)
    tot = np.zeros((p, d), f32)
    for t in range(12):
        tot = (tot + unit(draw(seed, gid, gen, base + 1 + t, TAG_MUT))).astype(f32)
    gauss = (tot / f32(12.0)).astype(f32)
    nx = (v + ((ek * s).astype(f32) * gauss).astype(f32)).astype(f32)
    bad = (nx < 0) | (nx > 1)
    gauss = np.where(bad, (gauss * f32(-0.5)).astype(f32), gauss)
    nx = np.where(bad, (v + ((ek * s).astype(f32) * gauss).astype(f32)).astype(f32), nx)
    es = np.exp((np.abs(gauss) - rtop).astype(f32).astype(np.float64))
    fac = np.power(ek.astype(np.float64), np.float64(beta)).astype(f32) * np.power(es.astype(f32).astype(np.float64), np.float64(bscale)).astype(f32)
    return nx.astype(f32), (s * fac.astype(f32)).astype(f32)


def main():
    tab = wavetable()
    out = {"wavetable": tab}
    cases = {
        0: [[1450 / 3520, 3 / 8, 200 / 3520, 1.0], [0.9, 1.0, 0.02, 0.5], [0.0007, 0.33, 0.005, 1.0]],
        1: [[3078 / 3520, 2 / 8, 3015 / 3520, 1.5 / 8, 3141 / 3520, 1 / 8], [0.2, 0.9, 0.6, 0.4, 0.1, 0.8],
            [1.0, 1.0, 1.0, 1.0, 1.0, 1.0]],
        2: [[0.41, 0.375, 0.057, 1.0, 0.2, 0.5, 0.11, 0.7, 0.6, 0.1, 0.3, 0.4],
            [0.05, 0.9, 0.8, 0.3, 0.5, 0.5, 0.5, 0.5, 0.95, 0.2, 0.01, 1.0]],
        3: [[0.3, 0.25, 0.85, 0.19, 0.89, 0.125, 0.5, 0.1], [0.7, 0.6, 0.5, 0.4, 0.3, 0.2, 0.1, 0.9]],
    }
    for n in (1024,):
        win, wf = window(n)
        out[f"window_{n}"] = win
        out[f"window_factor_{n}"] = np.array([wf], f32)
        for kind, plist in cases.items():
            params = np.array(plist, f32)
            audio = np.stack([synth(kind, pv, n, tab) for pv in params])
            mags = np.stack([spectrum(a, win, wf) for a in audio])
            fit = np.array([fitness(m, mags[0]) for m in mags], f32)
            out[f"k{kind}_n{n}_params"] = params
            out[f"k{kind}_n{n}_audio"] = audio
            out[f"k{kind}_n{n}_mag"] = mags
            out[f"k{kind}_n{n}_fitness_vs_row0"] = fit
    n = 4096
    win, wf = window(n)
    out[f"window_factor_{n}"] = np.array([wf], f32)
    params = np.array(cases[3][:1], f32)
    audio = np.stack([synth(3, pv, n, tab) for pv in params])
    out["k3_n4096_params"] = params
    out["k3_n4096_audio"] = audio
    out["k3_n4096_mag"] = np.stack([spectrum(a, win, wf) for a in audio])
    # non-zero parameter minimum (scaleParams, Evolutionary_Strategy.hpp:567-576)
    out["k0_pmin"] = np.array([100.0, 0.5, 50.0, 0.1], f32)
    out["k0_pmin_audio"] = synth(0, cases[0][1], 1024, tab, pmin=[100.0, 0.5, 50.0, 0.1])[None]

    seed, gid_base = 0x5EED0001, 4096
    v0, s0 = init_population(64, 6, seed, gid_base, 2)
    out["init_values"], out["init_steps"] = v0, s0
    rv, rs = recombine(v0, s0, 32, 16)
    out["recombine_values"], out["recombine_steps"] = rv, rs
    s_big = s0.copy()
    s_big[:8] = f32(2.5)
    mv, ms = mutate(rv, s_big, seed, gid_base, 9)
    out["mutate_in_steps"] = s_big
    out["mutate_values"], out["mutate_steps"] = mv, ms
    out["meta_seed_gid_chunk_parents_block_gen"] = np.array([seed, gid_base, 2, 32, 16, 9], np.uint64)

    rng = np.random.default_rng(42)
    fs = rng.random(200).astype(f32)
    fs[[3, 50, 120]] = fs[7]
    fs[[10, 199]] = np.nan
    fs[20] = np.inf
    fs[30], fs[31], fs[32] = 0.0, -0.0, 0.0
    out["sort_fitness"] = fs
    out["sort_perm"] = np.argsort(fs, kind="stable").astype(np.uint32)

    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    print("wrote golden_v1.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
