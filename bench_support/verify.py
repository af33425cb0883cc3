"""bench_support.verify -- the bench line vouches for its own numbers.

What a timed region wrote is downloaded AFTER the region (never inside it) together with the inputs it read, and compared byte
for byte with the oracle's answer for those inputs -- the reference's own kind of check (src/NativeUtils/main.cpp:211-245: a
bit-compare of the produced mesh with a stored one).  The oracle is the checker here, never the thing measured; a mismatch puts
"error" into the line and the run ends non-zero (bench.py)."""
import numpy as np


def compare_cloud(orc, depth_u16, rgb_u8, widths, heights, intr, wt, bounds, got_vertices_u8, got_offsets, n_threads=8):
    """One tick: the oracle's merge of (depth, rgb) against the cloud bytes [n, 16] and the offset table [N + 1] a run produced.
    Returns (ok, detail)."""
    want, counts = orc.generate_mesh_vertices(depth_u16, rgb_u8, widths, heights, intr, wt, bounds, n_threads=n_threads)
    want_off = np.concatenate([[0], np.cumsum(np.asarray(counts, np.int64))])
    got_off = np.asarray(got_offsets, np.int64).ravel()
    if got_off.shape != want_off.shape or not np.array_equal(got_off, want_off):
        return False, f"offsets {got_off.tolist()} != oracle {want_off.tolist()}"
    got = np.ascontiguousarray(got_vertices_u8, dtype=np.uint8).reshape(-1, 16)
    if len(got) != len(want):
        return False, f"{len(got)} vertices != oracle {len(want)}"
    wb = want.view(np.uint8).reshape(-1, 16)
    if got.tobytes() != wb.tobytes():
        bad = np.flatnonzero((got != wb).any(axis=1))
        return False, f"{len(bad)} of {len(want)} vertices differ, first at {int(bad[0])}"
    return True, f"{len(want)} vertices"


def compare_mesh(orc, depth_in_u16, rgb_in_u8, widths, heights, intr, wt, bounds, got_corr_depth_u16, got_corr_rgb_u8, got_vertices_u8,
                 got_offsets, got_triangles_i32, got_tri_offsets, n_threads=8):
    """One tick of the chained tick: corrected maps, vertices, offsets and triangles against the oracle's radial correction + mesh."""
    wd, wc = orc.radial_correction(depth_in_u16, rgb_in_u8, widths, heights, intr, n_threads=n_threads)
    wd = np.ascontiguousarray(np.asarray(wd)).view(np.uint8).ravel()
    wc = np.ascontiguousarray(np.asarray(wc)).ravel()
    if np.ascontiguousarray(got_corr_depth_u16).view(np.uint8).tobytes() != wd.tobytes():
        return False, "corrected depth maps differ"
    if np.ascontiguousarray(got_corr_rgb_u8).tobytes() != wc.tobytes():
        return False, "corrected colour maps differ"
    want_v, counts, want_t = orc.generate_mesh(wd, wc, widths, heights, intr, wt, bounds)
    want_off = np.concatenate([[0], np.cumsum(np.asarray(counts, np.int64).ravel()[:len(widths)])])
    if not np.array_equal(np.asarray(got_offsets, np.int64).ravel(), want_off):
        return False, "vertex offsets differ"
    got = np.ascontiguousarray(got_vertices_u8, dtype=np.uint8).reshape(-1, 16)
    if got.tobytes() != want_v.view(np.uint8).reshape(-1, 16).tobytes():
        return False, "vertices differ"
    want_t = np.asarray(want_t, np.int32).reshape(-1, 3)
    got_t = np.asarray(got_triangles_i32, np.int32).reshape(-1, 3)
    if int(np.asarray(got_tri_offsets).ravel()[-1]) != len(want_t) or got_t.shape != want_t.shape or not np.array_equal(got_t, want_t):
        return False, f"triangles differ ({len(got_t)} against the oracle's {len(want_t)})"
    return True, f"{len(want_v)} vertices, {len(want_t)} triangles"


def verify_clouds(torch, depth, rgb, vertices, offsets, ticks, widths, heights, intr, wt, bounds, n_threads=8):
    """Device tensors of a finished run: depth [T, P] (u16 bit patterns), rgb [T, 3P], vertices [T, capacity, 16] u8, offsets
    [T, N + 1]; `ticks` = the ticks to check.  Returns the record that goes into the line."""
    from oracle import orc
    torch.cuda.synchronize()
    ticks = sorted(set(int(t) for t in ticks))
    rec = {"ticks": ticks, "bitexact": True, "against": "oracle merge of the same frames (bytes of every vertex and the offset tables)", "vertices_checked": 0}
    for t in ticks:
        off = offsets[t].cpu().numpy()
        n = int(off[-1])
        ok, detail = compare_cloud(orc, depth[t].cpu().numpy().view(np.uint16), rgb[t].cpu().numpy(), widths, heights, intr, wt, bounds,
                                   vertices[t, :n].cpu().numpy(), off, n_threads=n_threads)
        if not ok:
            rec["bitexact"] = False
            rec["first_mismatch"] = f"tick {t}: {detail}"
            break
        rec["vertices_checked"] += n
    return rec


def verify_mesh_tick(torch, tick, d_in, c_in, d_corr, c_corr, vertices, offsets, triangles, tri_offsets, widths, heights, intr, wt, bounds, n_threads=8):
    """The chained tick's outputs of one tick (device tensors, [T, ...]) against the oracle."""
    from oracle import orc
    torch.cuda.synchronize()
    off, toff = offsets[tick].cpu().numpy(), tri_offsets[tick].cpu().numpy()
    nv, nt = int(off[-1]), int(toff[-1])
    ok, detail = compare_mesh(orc, d_in[tick].cpu().numpy().view(np.uint16), c_in[tick].cpu().numpy(), widths, heights, intr, wt, bounds,
                              d_corr[tick].cpu().numpy().view(np.uint16), c_corr[tick].cpu().numpy(), vertices[tick, :nv].cpu().numpy(), off,
                              triangles[tick, :nt].cpu().numpy(), toff, n_threads=n_threads)
    rec = {"ticks": [int(tick)], "bitexact": bool(ok), "against": "oracle radial correction + mesh of the same frames (corrected maps, vertices, offsets, triangles)",
           "checked": detail}
    if not ok:
        rec["first_mismatch"] = f"tick {tick}: {detail}"
    return rec


def checked(fn):
    """A check that could not be carried out (the oracle library is missing on the box, an exception on a path a first multi-GPU run takes for
    the first time) is not a mismatch: the record says what happened (`bitexact: null`, `not_checked`) and the measurement stands.  Only
    `bitexact: false` -- outputs that differ from the oracle -- fails the run."""
    def run(*a, **k):
        try:
            return fn(*a, **k)
        except Exception as ex:  # noqa: BLE001
            return {"bitexact": None, "not_checked": f"{type(ex).__name__}: {ex}"}
    return run


def failures(result):
    """Every value_verified record of the line whose outputs DIFFER from the oracle (bitexact false), as (path, record)."""
    out = []

    def walk(node, path):
        if isinstance(node, dict):
            v = node.get("value_verified")
            if isinstance(v, dict) and v.get("bitexact") is False:
                out.append((path + "value_verified", v))
            for k, x in node.items():
                if k != "value_verified":
                    walk(x, f"{path}{k}.")
    walk(result, "")
    return out
