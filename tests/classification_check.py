"""Per-(tile, triangle) check of the product's conservative classification against the reference's own arithmetic.

The reference scans every triangle for every ray (RayTracer/Kernels.cuh:75-92).  The product drops most (tile, triangle)
pairs on interval proofs (csrc/rt_trace.hpp: tile_misses_triangle), skips the rays of tiles whose farthest hit is certain,
and (large scenes) skips candidate tests per sample batch on the per-sample forms.  Comparing final IMAGES cannot see a
wrongly dropped triangle that is not the farthest hit (Kernels.cuh:73,84), so this module checks the verdicts themselves:

  device side   rt_dbg_classify (include/rt_mi355x.h): for a region of the frame -- an 8x8 wave tile, a 32x8 block, a
                128x64 macro tile, bounded exactly as the trace path bounds them -- and EVERY triangle: kept / certainly
                hit, and the interval ends det', U', V', q the verdict was taken from, at the product's rounding allowances
                (scale 1) and with every allowance scaled to 0.3 ... 0 (level 3: the 32x16 region of the small scenes'
                two-level list builder); rt_dbg_read_tile_lists: the lists and certain-winner verdicts the product stored;
  oracle side   orc_tile_probe (oracle/oracle.h): the reference's HitTriangle arithmetic and farthest-hit scan for the rays
                of every pixel of the region x a set of lens samples (rim of the lens, centre, the pixels' own RNG stream)
                against every triangle.

Checked per (region, triangle):
  (a) no probed ray hits a triangle the classification dropped;
  (b) every probed ray hits a triangle it called certainly hit; in a tile with a certain winner every ray's farthest hit
      (first-scanned on ties) is that triangle;
  (c) the reference's per-ray det |w|, U |w|, V |w| (w = focal point - lens point) lie inside the exported intervals, and
      t / |w| of the hit rays inside the exported q bounds;
  (d) (forms) no ray that the per-sample forms skip is hit.
MARGIN: how far the reference's values reach beyond the zero-allowance intervals, as a fraction of what the product charges
("needed scale": 0 = the allowances are never touched, 1 = fully used, > 1 = containment broken), and the smallest scale of
the ladder at which (a) and (b) still hold.

Test infrastructure only (tests/, tools/classification_margin.py)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

LADDER = (1000, 300, 100, 30, 10, 0)


def workers():
    n = len(os.sched_getaffinity(0))
    return max(1, min(16, n))


def lens_samples(orc, seed=1, pixel_index=0, n_rng=160, n_rim=72):
    """Points of the unit disk as random::UnifromOnDisk produces them (Random.cuh:13-19): the centre, the rim (sr = 1 at
    n_rim angles and sr one and a few ulps below 1), half radius, and the first n_rng samples of a pixel's own stream."""
    pts = [(0.0, 0.0)]
    one = np.float32(1.0)
    below = np.nextafter(one, np.float32(0.0))
    for k in range(n_rim):
        t = np.float32(2.0 * 3.14156545) * np.float32((k + 0.5) / n_rim)
        s, c = orc.sincos(t)
        for sr in ((one, below) if k % 3 == 0 else (one,)):
            pts.append((float(np.float32(sr * c)), float(np.float32(sr * s))))
        if k % 6 == 0:
            pts.append((float(np.float32(np.float32(0.5) * c)), float(np.float32(np.float32(0.5) * s))))
    for t in (0.0, 0.5 * np.pi, np.pi, 1.5 * np.pi):                         # the axes: where the lens box touches the disk
        s, c = orc.sincos(np.float32(t))
        pts.append((float(c), float(s)))
    st = orc.rng_init(seed, pixel_index)
    for _ in range(n_rng):
        xy = orc.uniform_on_disk(st)
        pts.append((float(xy[0]), float(xy[1])))
    return np.array(pts, np.float32)


def region_pixels(x0, y0, w, h, W, rows, row0, max_pixels=None, rng=None):
    """Full-frame (x, y) of the region's in-band pixels; above max_pixels: the border ring + a random interior sample."""
    xs = np.arange(x0, min(x0 + w, W))
    ys = np.arange(y0, min(y0 + h, rows))
    gx, gy = np.meshgrid(xs, ys)
    pix = np.stack([gx.ravel(), gy.ravel() + row0], axis=1).astype(np.uint32)
    if max_pixels is not None and pix.shape[0] > max_pixels:
        border = (gx.ravel() == xs[0]) | (gx.ravel() == xs[-1]) | (gy.ravel() == ys[0]) | (gy.ravel() == ys[-1])
        b = pix[border]
        if b.shape[0] > max_pixels // 2:                                       # corners + an even sample of the ring
            keep = np.unique(np.concatenate([np.linspace(0, b.shape[0] - 1, max_pixels // 2).astype(int)]))
            corners = pix[[0, len(xs) - 1, pix.shape[0] - len(xs), pix.shape[0] - 1]]
            b = np.concatenate([corners, b[keep]])
        inner = pix[~border]
        take = max(0, max_pixels - b.shape[0])
        if take and inner.shape[0]:
            sel = (rng or np.random.default_rng(0)).choice(inner.shape[0], min(take, inner.shape[0]), replace=False)
            pix = np.concatenate([b, inner[sel]])
        else:
            pix = b
    return pix


class Tally:
    """What a run found, summed over regions."""

    def __init__(self):
        self.regions = self.pairs = self.rays = 0
        self.dropped = self.kept = self.sure = self.sure_tiles = 0
        self.kept_hit = 0               # kept pairs that some probed ray really hits (how tight the classification is)
        self.bad = {s: {"drop_hit": 0, "drop_win": 0, "sure_miss": 0, "tile_winner": 0} for s in LADDER}
        self.scales = set()             # the scales of the ladder this tally was run at
        self.contain_bad = 0            # (c) at the product's allowances
        self.q_bad = 0
        self.form_wrong = 0
        self.form_rejects = self.form_tests = 0
        self.needed = 0.0               # max needed scale of the allowances: det', U', V'
        self.needed_by = {"det": 0.0, "U": 0.0, "V": 0.0, "S": 0.0}
        self.needed_q = 0.0             # how much of the 1e-4 margin of the winner rule the q bounds use
        self.nan_pairs = 0
        self.pair_tiles = 0             # certain-winner tiles the product proved with the pairwise bound (beyond the interval rule)
        self.examples = []

    def merge(self, o):
        for k in ("regions", "pairs", "rays", "dropped", "kept", "kept_hit", "sure", "sure_tiles", "contain_bad", "q_bad", "form_wrong",
                  "form_rejects", "form_tests", "nan_pairs", "pair_tiles"):
            setattr(self, k, getattr(self, k) + getattr(o, k))
        for s in LADDER:
            for k in self.bad[s]:
                self.bad[s][k] += o.bad[s][k]
        self.scales |= o.scales
        self.needed = max(self.needed, o.needed)
        self.needed_q = max(self.needed_q, o.needed_q)
        for k in self.needed_by:
            self.needed_by[k] = max(self.needed_by[k], o.needed_by[k])
        self.examples = (self.examples + o.examples)[:12]

    def smallest_passing_scale(self):
        """smallest scale of the ladder (as a fraction) down to which (a) and (b) hold without exception"""
        ok = 1.0
        for s in LADDER:
            if s not in self.scales:
                continue
            if any(self.bad[s].values()):
                break
            ok = s / 1000.0
        return ok

    def summary(self):
        return {"regions": self.regions, "tile_triangle_pairs": self.pairs, "rays_per_pair_total": self.rays,
                "dropped_pairs": self.dropped, "kept_pairs": self.kept, "kept_pairs_hit_by_a_probed_ray": self.kept_hit, "certainly_hit_pairs": self.sure,
                "certain_winner_tiles": self.sure_tiles, "of_those_by_the_pairwise_bound": self.pair_tiles,
                "violations_by_scale": {str(s / 1000.0): dict(self.bad[s]) for s in LADDER if s in self.scales},
                "smallest_passing_scale": self.smallest_passing_scale(),
                "containment_violations_at_scale_1": self.contain_bad, "q_violations_at_scale_1": self.q_bad,
                "needed_scale": round(self.needed, 5), "needed_scale_by_quantity": {k: round(v, 5) for k, v in self.needed_by.items()},
                "needed_share_of_q_margin": round(self.needed_q, 5),
                "forms": {"tests": self.form_tests, "rejected": self.form_rejects, "rejected_but_hit": self.form_wrong},
                "pairs_with_nan": self.nan_pairs, "examples": self.examples}


def _needed(vmin, vmax, lo1, hi1, lo0, hi0):
    """Per triangle: the share of the allowance (interval at scale 1 minus interval at scale 0) the values reach into."""
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        up = np.where(vmax > hi0, (vmax - hi0) / np.maximum(hi1 - hi0, 1e-300), 0.0)
        dn = np.where(vmin < lo0, (lo0 - vmin) / np.maximum(lo0 - lo1, 1e-300), 0.0)
    return np.maximum(up, dn)


SURE_MAX_TRIS = 256     # every small scene (tile_lists_kernel carries the verdict across classification steps)


def check_region(tag, probe, nohit, rays, recs, hdrs, forms=False, tile_word=None, stored_list=None):
    """One region: `recs`/`hdrs` = {scale: records (n_tris, 12|32) / header (16,)} from rt_dbg_classify, `probe` from
    orc_tile_probe over `rays` rays.  tile_word/stored_list: the product's stored tile list (small scenes, level 0)."""
    t = Tally()
    n = probe.shape[0]
    t.regions, t.pairs, t.rays = 1, n, rays * n
    r1, h1 = recs[1000].astype(np.float64), hdrs[1000]
    usable = (int(h1[8]) & 1) != 0
    assert int(h1[8]) & 2, "%s: the two focal-bound paths (list builder / trace wave) disagree" % tag
    flags1 = recs[1000][:, 0].astype(np.int32)
    keep1 = (flags1 & 1) != 0
    sure1 = (flags1 & 2) != 0
    t.kept, t.dropped, t.sure = int(keep1.sum()), int((~keep1).sum()), int(sure1.sum())
    hits, wins = probe["hits"], probe["wins"]
    real_hits = hits - probe["nan_hits"]                     # a hit with t = NaN never wins (Kernels.cuh:84: distance < NaN is false)
    t.kept_hit = int((keep1 & (hits > 0)).sum())
    t.nan_pairs = int((probe["nan_rays"] > 0).sum())
    for s in LADDER:
        if s not in recs:
            continue
        t.scales.add(s)
        fl = recs[s][:, 0].astype(np.int32)
        keep, sure = (fl & 1) != 0, (fl & 2) != 0
        b = t.bad[s]
        b["drop_hit"] = int(((~keep) & (real_hits > 0)).sum())
        b["drop_win"] = int(((~keep) & (wins > 0)).sum())
        b["sure_miss"] = int((sure & (hits < probe["rays"])).sum())
        if not forms:
            win = certain_winner(recs[s], (int(hdrs[s][8]) & 1) != 0)
            if win is not None and wins[win] != rays:
                b["tile_winner"] = 1
        if (b["drop_hit"] or b["sure_miss"] or b["tile_winner"]) and s == 1000:
            i = int(np.argmax(((~keep) & (real_hits > 0)) | (sure & (hits < probe["rays"]))))
            t.examples.append("%s: triangle %d flags %d hits %d/%d wins %d" % (tag, i, fl[i], hits[i], probe["rays"][i], wins[i]))
    if not forms:
        win = certain_winner(recs[1000], usable)
        if win is not None:
            t.sure_tiles = 1
        if tile_word is not None:
            # The PRODUCT's own list and verdict for this tile (rt_dbg_read_tile_lists).  The two-level builder keeps what
            # survives the region's family AND the tile's: a subset of the tile-level verdicts exported here, ascending.  The
            # checks that matter run on the product's list itself: nothing it dropped is hit, its winner wins every ray.
            cnt, pw, pflag = int(tile_word & 0x3FF), int((tile_word >> 10) & 0x3FF), bool(tile_word >> 31)
            lst = stored_list[:cnt].astype(np.int64)
            assert cnt <= n and (np.diff(lst) > 0).all() and (lst < n).all(), "%s: stored list not ascending / out of range: %s" % (tag, lst)
            pk = np.zeros(n, bool)
            pk[lst] = True
            assert not (pk & ~keep1).any(), "%s: the product keeps triangles the tile-level classification drops: %s vs %s" % (tag, lst, np.flatnonzero(keep1))
            b = t.bad[1000]
            b["drop_hit"] = int(((~pk) & (real_hits > 0)).sum())
            b["drop_win"] = int(((~pk) & (wins > 0)).sum())
            t.kept, t.dropped = int(pk.sum()), int((~pk).sum())
            rec_p = recs[1000].copy()
            rec_p[~pk, 0] = 0.0                                # the winner rule over the product's candidates
            hw = certain_winner(rec_p, usable) if n <= SURE_MAX_TRIS else None
            # the interval rule implies the product's verdict; the product may also order A's rivals pairwise (pair_farther,
            # not re-derived here): such a verdict is judged by the rays alone, like every other (tile_winner below)
            assert (hw is None or (pflag and pw == hw)), "%s: stored certain-winner verdict (%s, %d) vs harness %s" % (tag, pflag, pw, hw)
            if pflag and hw is None:
                t.pair_tiles += 1
                f1 = recs[1000][:, 0].astype(np.int32)
                assert (f1[pw] & 3) == 3, "%s: stored winner %d is not a kept, certainly-hit triangle of the tile-level verdicts" % (tag, pw)
            b["tile_winner"] = int(pflag and wins[pw] != rays)
            t.sure_tiles = int(pflag)
            if (b["drop_hit"] or b["tile_winner"]) and len(t.examples) < 12:
                t.examples.append("%s: product list %s flag %s winner %d: hits on dropped %s, winner wins %d of %d rays" % (
                    tag, lst.tolist(), pflag, pw, np.flatnonzero((~pk) & (real_hits > 0)).tolist(), wins[pw] if pflag else -1, rays))
    if usable:
        r0 = recs[0].astype(np.float64) if 0 in recs else None
        fin = np.isfinite(r1[:, 1:7]).all(axis=1) & (probe["nan_rays"] == 0)
        s_lo = 7 if forms else 9                              # S' = det' - U' - V' as one polynomial (the third-edge rules);
        fin_all = fin                                         # -inf / +inf where the instantiation leaves the S rules out
        for name, lo, hi, vmin, vmax in (("det", 1, 2, "det_min", "det_max"), ("U", 3, 4, "U_min", "U_max"), ("V", 5, 6, "V_min", "V_max"),
                                         ("S", s_lo, s_lo + 1, "S_min", "S_max")):
            fin = fin_all & np.isfinite(r1[:, lo]) & np.isfinite(r1[:, hi])
            bad = fin & ((probe[vmin] < r1[:, lo]) | (probe[vmax] > r1[:, hi]))
            t.contain_bad += int(bad.sum())
            if bad.any() and len(t.examples) < 12:
                i = int(np.argmax(bad))
                t.examples.append("%s: %s of triangle %d: reference [%.9g, %.9g] vs interval [%.9g, %.9g]" % (
                    tag, name, i, probe[vmin][i], probe[vmax][i], r1[i, lo], r1[i, hi]))
            if r0 is not None:
                f0 = fin & np.isfinite(r0[:, 1:7]).all(axis=1) & np.isfinite(r0[:, lo]) & np.isfinite(r0[:, hi])
                nd = _needed(probe[vmin], probe[vmax], r1[:, lo], r1[:, hi], r0[:, lo], r0[:, hi])
                nd = np.where(f0, nd, 0.0)
                t.needed_by[name] = float(nd.max(initial=0.0))
        t.needed = max(t.needed_by.values())
        if not forms:
            # q = t / |w|: upper bound over the rays that hit (any kept triangle), lower bound for certainly-hit ones.  The
            # winner rule keeps 1e-4 relative between the bounds it compares (two roundings of the quotient and of t itself).
            has = keep1 & (real_hits > 0) & fin
            qlo, qhi = r1[:, 7], r1[:, 8]
            with np.errstate(invalid="ignore", over="ignore"):
                over = np.where(has & np.isfinite(qhi), (probe["q_max"] - qhi) / np.maximum(np.abs(qhi), 1e-300), 0.0)
                under = np.where(has & sure1 & np.isfinite(qlo), (qlo - probe["q_min"]) / np.maximum(np.abs(qlo), 1e-300), 0.0)
            worst = float(max(over.max(initial=0.0), under.max(initial=0.0)))
            t.needed_q = max(0.0, worst) / 1e-4
            t.q_bad = int(((over > 1e-4) | (under > 1e-4)).sum())
    if forms:
        t.form_tests = int(probe["rays"][keep1].sum())
        t.form_rejects = int(probe["form_rejects"][keep1].sum())
        t.form_wrong = int(probe["form_wrong"][keep1].sum())
        if t.form_wrong and len(t.examples) < 12:
            i = int(np.argmax(np.where(keep1, probe["form_wrong"], 0)))
            t.examples.append("%s: forms skip %d hit rays of triangle %d" % (tag, probe["form_wrong"][i], i))
    return t


def certain_winner(rec, usable):
    """The trace kernel's certain-winner rule (rt_trace.hpp, after the one-step classification) on exported records:
    the kept, certainly hit triangle with the largest lower bound of q -- lowest index on ties -- wins if every other kept
    triangle's upper bound stays below it by 1e-4 relative.  None: no certain winner."""
    if not usable:
        return None
    fl = rec[:, 0].astype(np.int32)
    keep, sure = (fl & 1) != 0, (fl & 2) != 0
    cand = keep & sure
    if not cand.any():
        return None
    qlo = rec[:, 7].astype(np.float32)
    qhi = rec[:, 8].astype(np.float32)
    Q = np.max(np.where(cand, qlo, -np.inf).astype(np.float32))
    best = np.flatnonzero(cand & (qlo == Q))
    if best.size == 0:                                        # Q is NaN
        return None
    A = int(best[0])
    other = keep.copy()
    other[A] = False
    if not other.any():
        return A
    qh = np.where(np.isnan(qhi), np.inf, qhi)
    R = np.float32(np.max(np.where(other, qh, -np.inf)))
    with np.errstate(invalid="ignore", over="ignore"):
        ok = R < np.float32(Q - np.float32(1e-4) * np.float32(np.abs(R) + np.abs(Q)))
    return A if bool(ok) else None


def run(g, o, regions, level, lens, *, forms=False, ladder=LADDER, max_pixels=None, stored=None, tag="", pool=None, row0=0):
    """Check `regions` [(x0, y0) band-local] of tracer `g` (api.RayTracer) against oracle tracer `o` (same scene, camera,
    frame).  stored: g.DebugTileListWords() to cross-check the product's own lists (level 0, small scenes).  -> Tally"""
    regions = np.ascontiguousarray(regions, np.uint32).reshape(-1, 2)
    rw, rh = {0: (8, 8), 1: (32, 8), 2: (128, 64), 3: (32, 16), 4: (512, 256)}[level]      # 4: a super tile of 4 x 4 macro tiles
    out = {s: g.DebugClassify(regions, level, forms, s) for s in ladder}
    if level == 3 and not (int(out[ladder[0]][0][0][8]) & 4):
        return Tally()                                      # the two-level list builder is not in use for this camera / frame
    rng = np.random.default_rng(7)
    pixsets = [region_pixels(int(x0), int(y0), rw, rh, g.width, g.rows, row0, max_pixels, rng) for x0, y0 in regions]

    def one(i):
        hdrs = {s: out[s][0][i] for s in ladder}
        recs = {s: out[s][1][i] for s in ladder}
        fo = recs[1000][:, 12:30] if forms else None
        probe, nohit = o.tile_probe(pixsets[i], lens, forms=fo, fc=hdrs[1000][13:16] if forms else None)
        rays = pixsets[i].shape[0] * lens.shape[0]
        word = lst = None
        if stored is not None:
            ty, tx = int(regions[i][1]) // 8, int(regions[i][0]) // 8
            word, lst = int(stored[ty, tx, 0]), stored[ty, tx, 1:]
        return check_region("%s level %d region (%d, %d)" % (tag, level, regions[i][0], regions[i][1]), probe, nohit, rays,
                            recs, hdrs, forms, word, lst)

    total = Tally()
    if pool is None:
        with ThreadPoolExecutor(workers()) as ex:
            parts = list(ex.map(one, range(regions.shape[0])))
    else:
        parts = list(pool.map(one, range(regions.shape[0])))
    for p in parts:
        total.merge(p)
    return total
