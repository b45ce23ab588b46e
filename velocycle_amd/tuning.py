"""`Tuning`: the engine's tuning as DATA (include/velocycle_hip.h: vc_tuning) -- SURVEY.md section 5's "engine config = plain C
struct mirrored by a Python dataclass".  Neither the library nor the package reads tuning from the environment: a `Tuning` is
handed to `HipEngine(..., tuning=)` / `fit(..., tuning=)`; the default object selects the measured defaults of DESIGN.md.

The ONE place where `VC_*` environment variables are turned into a Tuning is `Tuning.from_env()`, called explicitly by the A/B
scripts under profiles/tools and by bench.py (so that `VC_GPL=4 python bench.py` still measures a variant) -- never implicitly.
Every rank of a sharded run must use the same Tuning: `SVIRunner` compares `digest()` over the ranks before the first step.
"""
from __future__ import annotations

import ctypes as C
import hashlib
from dataclasses import dataclass, field, fields
from typing import Optional, Tuple

_HIST = {None: 0, "auto": 0, "lists": 1, "dense": 2}
_PW = {None: 0, "auto": 0, "off": 1, "force": 2}
_LAYOUT = {None: 0, "batched": 0, "serial": 1, "streams": 2}
_STORAGE = {None: 0, "auto": 0, "f32": 1}


@dataclass(frozen=True)
class Tuning:
    # ---- vc_tuning (the library) -----------------------------------------------------------------------------------
    genes_per_lane: int = 0                     # 0 auto | 4 | 8
    blocks_per_cu: int = 0                      # 0 = occupancy of the code object
    cells_per_wave: int = 0                     # 0 = balanced resident round
    pass_min_cw: int = 0                        # 0 = 12
    pass_shares: Optional[Tuple[float, ...]] = None   # None: measured defaults | (1,): equal shares | 2..4 shares
    tail_cells: int = 0                         # 0 auto | 256 | 512 | 1024
    count_storage: Optional[str] = None         # None / "auto" | "f32"
    host_hist: bool = False
    hist_dense: Optional[str] = None            # None / "auto" | "lists" | "dense"
    pw_inline: Optional[str] = None             # None / "auto" | "off" | "force"
    tail2: bool = True                          # False: three launches where two are the default
    tail_merged: bool = True                    # False: the tutorial flow without its merged second launch
    force_generic: bool = False
    particles_layout: Optional[str] = None      # None / "batched" | "serial" | "streams"
    dense_batches: bool = False                 # True: batch offsets as the dense Db contraction even for a one-hot Db
    p2p_timeout_s: float = 0.0                  # 0 = 2 s
    no_tail_spec: bool = False                  # True: the run-time-flag small kernels even where a compiled signature matches
    pw_lane: bool = True                        # False: the U-only kernel's wave-level reduction even where the per-lane nu_omega partials apply
    p2p_one_launch: bool = False                # True (exchange="p2p"): phases A and B in ONE launch, exchange at block granularity (opt-in)
    p2p_fold: bool = True                       # False: the peer-to-peer exchange as a launch of its own between phases A and B (rounds 3-5)
    # ---- the host side (SVIRunner / fit) ---------------------------------------------------------------------------
    adam_impl: Optional[str] = None             # single-rank perf step: None = "fused3"
    adam_impl_dist: Optional[str] = None        # sharded perf step: None = "sharded"
    exchange: Optional[str] = None              # who sums the exchange buffer: None = rule of SVIRunner | engine | torch | p2p | none
    exchange_check: bool = False                # self-check of the engine-owned exchange on a 1-rank group too
    particles_host_loop: bool = False           # K particles through the host loop instead of vc_svi_run_particles
    run_deadline_s: float = 900.0               # bounded wait of sharded runs

    def to_c(self):
        from . import _lib
        t = _lib.vc_tuning()
        t.genes_per_lane, t.blocks_per_cu = int(self.genes_per_lane), int(self.blocks_per_cu)
        t.cells_per_wave, t.pass_min_cw = int(self.cells_per_wave), int(self.pass_min_cw)
        sh = tuple(self.pass_shares) if self.pass_shares is not None else ()
        if len(sh) > 4:
            raise ValueError("at most four pass shares")
        t.n_pass_shares = len(sh)
        for i, x in enumerate(sh):
            t.pass_shares[i] = float(x)
        t.tail_cells = int(self.tail_cells)
        t.count_storage = _STORAGE[self.count_storage]
        t.host_hist = int(bool(self.host_hist))
        t.hist_dense = _HIST[self.hist_dense]
        t.pw_inline = _PW[self.pw_inline]
        t.no_tail2, t.no_tail_merged = int(not self.tail2), int(not self.tail_merged)
        t.force_generic = int(bool(self.force_generic))
        t.particles_layout = _LAYOUT[self.particles_layout]
        t.dense_batches = int(bool(self.dense_batches))
        t.p2p_timeout_s = float(self.p2p_timeout_s)
        t.no_tail_spec = int(bool(self.no_tail_spec))
        t.no_pw_lane = int(not self.pw_lane)
        t.p2p_separate = int(not self.p2p_fold)
        t.p2p_one_launch = int(bool(self.p2p_one_launch))
        return t

    def digest(self) -> int:
        """63-bit hash of every field: what the ranks of a sharded run compare (one MIN and one MAX all-reduce)."""
        h = hashlib.sha256(repr(tuple((f.name, getattr(self, f.name)) for f in fields(self))).encode()).digest()
        return int.from_bytes(h[:8], "little") >> 1

    def replace(self, **kw) -> "Tuning":
        import dataclasses
        return dataclasses.replace(self, **kw)

    @classmethod
    def from_env(cls, env=None) -> "Tuning":
        """The tuning the `VC_*` variables of `env` (default: os.environ) describe -- for A/B scripts and bench.py ONLY; nothing in
        the package calls this implicitly."""
        import os
        e = os.environ if env is None else env
        kw = {}
        geti = lambda k: int(e[k]) if e.get(k, "") != "" else 0
        kw["genes_per_lane"] = geti("VC_GPL")
        kw["blocks_per_cu"] = geti("VC_BLOCKS_PER_CU")
        kw["cells_per_wave"] = geti("VC_CELLS_PER_WAVE")
        kw["pass_min_cw"] = geti("VC_PASS_MIN_CW")
        if e.get("VC_PASS_SHARES"):
            parts = [float(x) for x in e["VC_PASS_SHARES"].replace(",", ":").split(":") if x != ""]
            kw["pass_shares"] = tuple(parts)
        kw["tail_cells"] = geti("VC_TAIL_TC")
        if e.get("VC_COUNT_STORAGE") == "f32":
            kw["count_storage"] = "f32"
        kw["host_hist"] = e.get("VC_HOST_HIST", "0") not in ("", "0")
        if e.get("VC_HIST_DENSE", "") != "":
            kw["hist_dense"] = "dense" if int(e["VC_HIST_DENSE"]) else "lists"
        if e.get("VC_PW_INLINE", "") != "":
            kw["pw_inline"] = {0: "off", 2: "force"}.get(int(e["VC_PW_INLINE"]), "auto")
        kw["tail2"] = e.get("VC_TAIL2", "1") != "0"
        kw["tail_merged"] = e.get("VC_TAIL_MERGED", "1") != "0"
        kw["force_generic"] = e.get("VC_FORCE_GENERIC", "0") not in ("", "0")
        if e.get("VC_PARTICLES_LAYOUT"):
            kw["particles_layout"] = e["VC_PARTICLES_LAYOUT"]
        kw["dense_batches"] = e.get("VC_DENSE_BATCHES", "0") not in ("", "0")
        kw["no_tail_spec"] = e.get("VC_NO_TAIL_SPEC", "0") not in ("", "0")
        kw["pw_lane"] = e.get("VC_PW_LANE", "1") != "0"
        kw["p2p_fold"] = e.get("VC_P2P_FOLD", "1") != "0"
        kw["p2p_one_launch"] = e.get("VC_P2P_ONE_LAUNCH", "0") not in ("", "0")
        if e.get("VC_P2P_TIMEOUT_S"):
            kw["p2p_timeout_s"] = float(e["VC_P2P_TIMEOUT_S"])
        kw["adam_impl"] = e.get("VC_ADAM_IMPL") or None
        kw["adam_impl_dist"] = e.get("VC_ADAM_IMPL_DIST") or None
        kw["exchange"] = e.get("VC_EXCHANGE") or None
        kw["exchange_check"] = e.get("VC_EXCHANGE_CHECK") == "1"
        kw["particles_host_loop"] = e.get("VC_PARTICLES_HOST_LOOP") == "1"
        if e.get("VC_RUN_DEADLINE_S"):
            kw["run_deadline_s"] = float(e["VC_RUN_DEADLINE_S"])
        return cls(**kw)
