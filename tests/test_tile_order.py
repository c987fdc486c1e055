"""CPU restatement of k_gemm_dma's workgroup -> tile mapping (csrc/sdn_gemm.hip: the XCD-chunked linear order followed by the
panel order of wide-N GEMMs) and the check that it is a bijection onto the tile grid for every grid shape the plans can produce --
a tile computed twice or never would not show up as a launch error.  The formulas below are the kernel's, line for line."""
import re
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tile_of(vid, tiles_m, tiles_n, panel):
    nt = tiles_m * tiles_n
    q, r, x = nt >> 3, nt & 7, vid & 7
    tile = (x * (q + 1) if x < r else r * (q + 1) + (x - r) * q) + (vid >> 3)
    tm, tn = divmod(tile, tiles_n)
    if panel:
        per_blk = 8 * tiles_n
        blk, rem = divmod(tile, per_blk)
        rows = min(8, tiles_m - blk * 8)
        full = rows * panel
        p, r2 = divmod(rem, full)
        w = min(panel, tiles_n - p * panel)
        tl = r2 // w
        tm, tn = blk * 8 + tl, p * panel + (r2 - tl * w)
    return tm, tn


def test_kernel_source_still_has_the_formulas_restated_here():
    src = open(os.path.join(ROOT, "safe_denoiser_amd", "csrc", "sdn_gemm.hip")).read()
    for frag in ("const int per_blk = 8 * g.tiles_n;", "const int rows = min(8, g.tiles_m - blk * 8);", "const int full = rows * g.panel;",
                 "const int p = rem / full, r2 = rem - p * full;", "const int w = min(g.panel, g.tiles_n - p * g.panel);",
                 "tm = blk * 8 + tl; tn = p * g.panel + (r2 - tl * w);", "g.panel = (g.tiles_n > 4 && g_gemm_variant != 15) ? 4 : 0;"):
        assert frag in src, frag
    assert re.search(r"tile = \(x < r \? x \* \(q \+ 1\) : r \* \(q \+ 1\) \+ \(x - r\) \* q\) \+ \(vid >> 3\);", src)


def test_every_tile_is_visited_exactly_once():
    for tiles_m in list(range(1, 41)) + [96, 195, 768, 771]:
        for tiles_n in range(1, 34):
            panel = 4 if tiles_n > 4 else 0
            nt = tiles_m * tiles_n
            seen = {tile_of(v, tiles_m, tiles_n, panel) for v in range(nt)}
            assert len(seen) == nt, (tiles_m, tiles_n)
            assert all(0 <= tm < tiles_m and 0 <= tn < tiles_n for tm, tn in seen), (tiles_m, tiles_n)


def test_a_wave_of_one_xcd_is_eight_rows_by_four_columns():
    """What the order is for: the 32 consecutive workgroups of one XCD cover 8 A row-tiles x 4 W column-tiles (row-major they cover
    2 rows x 16 columns at N = 5120)."""
    tiles_m, tiles_n = 768, 16                                        # GEGLU at C = 640, B = 192
    vids = [8 * j + 3 for j in range(32)]                             # XCD 3's first 32 workgroups
    tiles = [tile_of(v, tiles_m, tiles_n, 4) for v in vids]
    assert len({tm for tm, _ in tiles}) == 8 and len({tn for _, tn in tiles}) == 4
    rm = [tile_of(v, tiles_m, tiles_n, 0) for v in vids]
    assert len({tm for tm, _ in rm}) == 2 and len({tn for _, tn in rm}) == 16
