"""CPU model of the window / history scheme of csrc/mrfh_kernels.hpp (index logic only: one channel, float64, plain
convs): a block walks its share of the B x T output columns in windows of W columns; every conv's input image keeps the
rows in front of the NEXT window's column 0 in a history slot; only a run's first window starts `halo` columns early.
Checks the result against the direct computation for shares, windows and utterance ends of every shape."""
import numpy as np

FM = 32


def conv(img_rows, w, d, P, col0, n):
    """out[c] = sum_j w[j] img[FM + c - P + j d] for c in [col0, col0 + n)"""
    k = len(w)
    out = np.zeros(n)
    for j in range(k):
        out += w[j] * img_rows[FM + col0 - P + j * d: FM + col0 - P + j * d + n]
    return out


def lrelu(v, s=0.1):
    return np.maximum(v, v * s)


def direct(x, ws, ks, dils):
    rs = []
    T = len(x)
    for j in range(3):
        r = x.copy()
        for p in range(3):
            k, d = ks[j], dils[p]
            P1, P2 = (k - 1) * d // 2, (k - 1) // 2
            w1, w2 = ws[3 * j + p]
            a = np.concatenate([np.zeros(P1), lrelu(r), np.zeros(P1)])
            mid = sum(w1[i] * a[i * d: i * d + T] for i in range(k)) + 0.25
            a = np.concatenate([np.zeros(P2), lrelu(mid), np.zeros(P2)])
            r = sum(w2[i] * a[i: i + T] for i in range(k)) - 0.125 + r
        rs.append(r)
    return ((rs[0] + rs[1]) + rs[2]) / 3.0


def run(xs, ws, ks, dils, W, nblk, fold=None):
    B, T = xs.shape
    halo = max(sum((k - 1) * d // 2 + (k - 1) // 2 for d in dils) for k in ks)
    ol = 3 if fold is not None else 0
    vcols = W - halo
    adv = vcols - 2 * ol
    total = B * T
    out = np.full((B, T), np.nan)
    for share in range(nblk):
        g_lo, g_hi = total * share // nblk, total * (share + 1) // nblk
        if g_lo >= g_hi:
            continue
        ximg = np.zeros(W + 2 * FM)
        mimg = np.zeros(W + 2 * FM)
        hist_x = [np.zeros(25) for _ in range(9)]
        hist_m = [np.zeros(5) for _ in range(9)]

        def first(g):
            b = g // T
            a = g - b * T
            run_end = min(T, g_hi - b * T)
            tw = a - halo - ol
            return dict(b=b, tw=tw, lo=a, hi=min(run_end, tw + vcols - ol), run_end=run_end)
        it = first(g_lo)
        cols = np.arange(W)

        def window(at):
            t = at["tw"] + cols
            ok = (t >= 0) & (t < T)
            v = np.zeros(W)
            v[ok] = xs[at["b"], t[ok]]
            return v
        x0 = window(it)
        ximg[FM:FM + W] = lrelu(x0)
        while True:
            cur = dict(it)
            # next
            if it["hi"] < it["run_end"]:
                it = dict(it, tw=it["tw"] + adv, lo=it["hi"], hi=min(it["run_end"], it["tw"] + adv + vcols - ol))
                more = True
            else:
                g = it["b"] * T + it["run_end"]
                more = g < g_hi
                if more:
                    it = first(g)
            t = cur["tw"] + cols
            ok = (t >= 0) & (t < T)
            acc = None
            for j in range(3):
                xr = x0.copy()
                if j == 2 and more:
                    x0 = window(it)
                for p in range(3):
                    q = 3 * j + p
                    k, d = ks[j], dils[p]
                    P1, P2 = (k - 1) * d // 2, (k - 1) // 2
                    w1, w2 = ws[q]
                    # phase 1: conv1, save x rows, restore mid rows
                    c1 = conv(ximg, w1, d, P1, 0, W) + 0.25
                    hist_x[q][:P1] = ximg[FM + adv - P1: FM + adv]
                    mid = np.where(ok, lrelu(c1), 0.0)
                    mimg[FM:FM + W] = mid
                    mimg[FM - P2:FM] = hist_m[q][:P2]
                    # phase 2: conv2, save mid rows, restore next pair's x rows
                    c2 = conv(mimg, w2, 1, P2, 0, W) - 0.125
                    hist_m[q][:P2] = mimg[FM + adv - P2: FM + adv]
                    xr = np.where(ok, xr + c2, 0.0)
                    if p < 2:
                        ximg[FM:FM + W] = lrelu(xr)
                        qn, pn = q + 1, (ks[j] - 1) * dils[p + 1] // 2
                    else:
                        if j < 2 or more:
                            ximg[FM:FM + W] = lrelu(x0)
                        qn = 0 if j == 2 else q + 1
                        pn = (ks[(j + 1) % 3] - 1) * dils[0] // 2
                    ximg[FM - pn:FM] = hist_x[qn][:pn]
                acc = xr if j == 0 else acc + xr
            res = acc / 3.0
            if fold is not None:
                sb = np.where(ok, np.maximum(res, 0.01 * res), 0.0)
                for c0 in range(adv):
                    col = c0 + ol
                    tt = cur["tw"] + col
                    if cur["lo"] <= tt < cur["hi"]:
                        out[cur["b"], tt] = np.tanh(sum(fold[jj] * sb[col - 3 + jj] for jj in range(7)) + 0.5)
            else:
                sel = (t >= cur["lo"]) & (t < cur["hi"])
                assert np.all(np.isnan(out[cur["b"], t[sel]]))
                out[cur["b"], t[sel]] = res[sel]
            if not more:
                break
    return out


if __name__ == "__main__":
    rng = np.random.RandomState(0)
    dils = (1, 3, 5)
    for ks, B, T, W, nblk, fold in [((3, 7, 11), 1, 40, 576, 1, False), ((3, 7, 11), 2, 200, 576, 4, False),
                                    ((3, 7, 11), 1, 1500, 576, 1, False), ((11, 3, 7), 3, 1201, 576, 2, False),
                                    ((7, 7, 3), 2, 2000, 576, 3, False), ((3, 7, 11), 1, 4003, 512, 5, False),
                                    ((3, 7, 11), 1, 1531, 576, 2, True), ((3, 7, 11), 2, 777, 512, 3, True),
                                    ((3, 7, 11), 1, 37, 576, 1, True), ((11, 11, 11), 2, 1100, 512, 1, False)]:
        ws = [(rng.randn(ks[q // 3]) / np.sqrt(ks[q // 3]), rng.randn(ks[q // 3]) / np.sqrt(ks[q // 3])) for q in range(9)]
        xs = rng.randn(B, T)
        fw = rng.randn(7) / 3 if fold else None
        got = run(xs, ws, ks, dils, W, nblk, fw)
        for b in range(B):
            ref = direct(xs[b], ws, ks, dils)
            if fold:
                a = np.concatenate([np.zeros(3), np.maximum(ref, 0.01 * ref), np.zeros(3)])
                ref = np.tanh(sum(fw[j] * a[j: j + T] for j in range(7)) + 0.5)
            err = np.abs(got[b] - ref).max()
            assert not np.isnan(got).any() and err < 1e-10, (ks, B, T, W, nblk, fold, b, err)
        print("ok", ks, B, T, W, nblk, fold)
