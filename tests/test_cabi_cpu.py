"""CPU: the C-ABI shared library loads without a GPU and exports every symbol include/velocycle_hip.h
declares; argument validation that needs no device memory works; the product has no CPU fallback."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from velocycle_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib, _lib.load()


def test_library_exports_every_declared_symbol():
    mod, lib = _lib()
    hdr = open(os.path.join(ROOT, "include", "velocycle_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vc_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes parsed"
    assert declared == set(mod.EXPORTS), declared ^ set(mod.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.vc_abi_version() == mod.VC_ABI_VERSION


def test_struct_sizes_match_header_layout():
    mod, _ = _lib()
    assert C.sizeof(mod.vc_config) == 12 * 4 + 4 * 8 + 8 * 4
    assert C.sizeof(mod.vc_layout) == 8 * (4 + 2 * mod.VC_P_COUNT + 2 + 2 * mod.VC_E_COUNT)
    assert C.sizeof(mod.vc_stats) == 4 * 8 + 2 * 4 + 96 + 2 * 8 + 4 * 4 + 4 * 4 + 2 * 4 + 32 + 2 * 4      # (+ tail_spec, tail_spec_matched, tail_spec_name, pw_lane, hist_split: ABI version 2)


def test_create_validates_and_reports_errors():
    mod, lib = _lib()
    h = C.c_void_p()
    cfg = mod.vc_config(abi_version=mod.VC_ABI_VERSION, model=1, guide=0, noise=0, with_delta_nu=0, n_harmonics=80,
                        n_harmonics_w=1, Nb=1, Nx=1, lrmn_rank=5, rank=0, world_size=1, Ng=10, Nc_local=10,
                        Nc_global=10, cell_offset=0, gamma_alpha=1, gamma_beta=2, sigma_ln_s=.1, sigma_ln_u=.1,
                        rho_mean=4, rho_std=1, rho_scale=1)
    # 80 harmonics: 161 rows of per-gene state per wave, beyond the LDS (7 harmonics, refused until round 3, run on the
    # run-time-sized kernel set since round 4)
    assert lib.vc_create(C.byref(cfg), C.byref(h)) == mod.VC_ERR_UNSUPPORTED
    assert b"n_harmonics" in lib.vc_last_error(None)
    cfg.n_harmonics = 7
    assert lib.vc_create(C.byref(cfg), C.byref(h)) == mod.VC_OK
    lib.vc_destroy(h)
    cfg.n_harmonics = 1
    cfg.abi_version = 99
    assert lib.vc_create(C.byref(cfg), C.byref(h)) == mod.VC_ERR_ARG
    cfg.abi_version = mod.VC_ABI_VERSION
    cfg.Ng = 0
    assert lib.vc_create(C.byref(cfg), C.byref(h)) == mod.VC_ERR_ARG
    # a valid config creates an engine and a layout without touching the GPU
    cfg.Ng = 10
    assert lib.vc_create(C.byref(cfg), C.byref(h)) == mod.VC_OK
    lay = mod.vc_layout()
    assert lib.vc_get_layout(h, C.byref(lay)) == mod.VC_OK
    assert lay.header == 4 and lay.n_local == 20
    # mean-field velocity, NB, Nx=1, Hw=1: nu (2*30) + logbeta (2*10) + loggamma (2*10) + nuw (2*3) + shape_inv 10
    assert lay.n_global == 60 + 20 + 20 + 6 + 10 and lay.total == 4 + lay.n_global + 20
    assert lay.eps_n_global == 10 + 10 + 30 + 3 + 1 and lay.eps_total == lay.eps_n_global + 20    # global block padded to even
    # call-order errors come back as codes + messages, never as crashes
    assert lib.vc_elbo_grad(h, None, None, 0, 0, None, None, None, 1, None) == mod.VC_ERR_STATE
    assert b"before vc_finalize" in lib.vc_last_error(h)
    assert lib.vc_set_prior(h, 0, None, 3) == mod.VC_ERR_ARG
    lib.vc_destroy(h)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    from tests import helpers as H
    from velocycle_amd.engine import HipEngine, HipEngineError
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_phase_nb.npz")
    with pytest.raises(HipEngineError, match="no CPU fallback"):
        HipEngine(H.spec_from_fixture(z))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "velocycle_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f


def test_tuning_is_data_not_environment(monkeypatch):
    """the library reads no environment variable (no getenv in csrc/), the package reads none for tuning outside
    Tuning.from_env(), the ctypes mirror of vc_tuning has the header's layout, and the defaults are all-zero."""
    import ctypes as C
    import glob
    import re
    from velocycle_amd import _lib
    from velocycle_amd.tuning import Tuning
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in glob.glob(os.path.join(root, "velocycle_amd", "csrc", "*.h*")):
        assert "getenv" not in open(f).read(), f
    for f in glob.glob(os.path.join(root, "velocycle_amd", "*.py")):
        src = open(f).read()
        if os.path.basename(f) in ("tuning.py", "_lib.py"):
            continue
        assert "os.environ" not in src and "getenv" not in src, f
    hdr = open(os.path.join(root, "include", "velocycle_hip.h")).read()
    body = hdr[hdr.index("typedef struct vc_tuning {"):hdr.index("} vc_tuning;")]
    names = re.findall(r"^\s*(?:int32_t|float)\s+(\w+)(?:\[\d+\])?;", body, flags=re.M)
    assert names == [n for n, _ in _lib.vc_tuning._fields_], (names, [n for n, _ in _lib.vc_tuning._fields_])
    assert C.sizeof(_lib.vc_tuning) == 4 * (5 + 4 + 10 + 1 + 7)
    assert bytes(Tuning().to_c()) == bytes(C.sizeof(_lib.vc_tuning))
    # from_env is explicit and complete: every knob the A/B scripts used to export
    env = {"VC_GPL": "4", "VC_PASS_SHARES": "0.5:0.3:0.2", "VC_TAIL_TC": "512", "VC_COUNT_STORAGE": "f32", "VC_HIST_DENSE": "0",
           "VC_PW_INLINE": "2", "VC_TAIL2": "0", "VC_FORCE_GENERIC": "1", "VC_PARTICLES_LAYOUT": "streams", "VC_EXCHANGE": "p2p",
           "VC_RUN_DEADLINE_S": "3", "VC_ADAM_IMPL_DIST": "hip", "VC_P2P_TIMEOUT_S": "1.5", "VC_DENSE_BATCHES": "1"}
    t = Tuning.from_env(env)
    assert (t.genes_per_lane, t.pass_shares, t.tail_cells, t.count_storage, t.hist_dense, t.pw_inline, t.tail2, t.force_generic,
            t.particles_layout, t.exchange, t.run_deadline_s, t.adam_impl_dist, t.p2p_timeout_s, t.dense_batches) == \
        (4, (0.5, 0.3, 0.2), 512, "f32", "lists", "force", False, True, "streams", "p2p", 3.0, "hip", 1.5, True)
    c = t.to_c()
    assert (c.genes_per_lane, c.n_pass_shares, c.tail_cells, c.count_storage, c.hist_dense, c.pw_inline, c.no_tail2, c.force_generic,
            c.particles_layout, c.dense_batches) == (4, 3, 512, 1, 1, 2, 1, 1, 2, 1)
    assert Tuning.from_env({}) == Tuning() and Tuning().digest() != t.digest() and t.digest() == Tuning.from_env(env).digest()
    # an ambient variable changes nothing by itself
    monkeypatch.setenv("VC_GPL", "4")
    assert Tuning() == Tuning.from_env({})


def test_compiled_signature_rows_are_well_formed():
    """csrc/vc_tail_spec_rows.inc (the configurations the small kernels are compiled for, printed by profiles/tools/print_signature.py):
    27 ints per signature in the order of VC_SIG_FIELDS, launch kinds a bit set of {1, 2, 4, 8}, MQ one of the gene blocks' row bounds
    and >= the row count nq, names unique, -1 (left open) only on the fields a "multi" row opens -- and a row that leaves pw_inline
    open stands BEHIND every rank row (pw_inline = 0) it would otherwise shadow (vc_spec_match takes the first row that fits)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = open(os.path.join(root, "velocycle_amd", "csrc", "vc_common.h")).read()
    block = common[common.index("#define VC_SIG_FIELDS(X)"):common.index("struct VcSig {")]
    fields = re.findall(r"X\((\w+)\)", block) + ["cond"]
    assert len(fields) == 27 and "#define VC_SIG_INTS 27" in common
    rows = []
    for line in open(os.path.join(root, "velocycle_amd", "csrc", "vc_tail_spec_rows.inc")):
        m = re.match(r'\s*\{"(\w+)", (\d+), (\d+), \{(.*)\}\},', line)
        if m:
            vals = [int(x.rstrip("u")) for x in m.group(4).split(",")]
            rows.append((m.group(1), int(m.group(2)), int(m.group(3)), dict(zip(fields, vals)), len(vals)))
    assert len(rows) >= 6 and len({r[0] for r in rows}) == len(rows)
    for name, kinds, mq, sig, n in rows:
        assert n == 27 and 1 <= kinds <= 15 and mq in (2, 4, 6, 14) and mq >= sig["nq"] > 0, name
        opened = {f for f, v in sig.items() if v == -1}
        assert opened <= {"Nb", "Nx", "NW", "K", "pw_inline"}, (name, opened)
        if opened:
            assert sig["onehot"] == 1 and sig["with_dnu"] == 1, name
        assert not (kinds & 4 and kinds & 3 and sig["pw_inline"] != 0), name        # a rank's kernels never take K_main's own partials
    for i, (name, kinds, mq, sig, n) in enumerate(rows):
        if sig["pw_inline"] == -1:
            twin = dict(sig, pw_inline=0)
            later = [r[0] for r in rows[i + 1:] if r[3] == twin]
            assert not later, (name, "shadows", later)
