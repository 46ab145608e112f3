"""CPU: the C-ABI library is built, loads, and exports every symbol include/alignnet_hip.h
declares.  No compute calls here (there is no GPU and no CPU fallback)."""
import os
import re
import sys

import pytest

import alignnet3d
from alignnet3d import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "alignnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(alignnet_[a-z0-9_]+)\s*\(", text)))


def test_library_built_and_exports_every_declared_symbol():
    lib = alignnet3d.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/alignnet_hip.h but not exported"
    assert sorted(_capi.SYMBOLS) == declared, "ctypes table out of sync with the header"
    assert lib.alignnet_abi_version() == _capi.ABI_VERSION


def test_struct_layout_matches_header():
    import ctypes as C
    # alignnet_widths = int32 n + int32[8]; config field order is asserted by size arithmetic
    assert C.sizeof(_capi.Widths) == 4 * 9
    assert C.sizeof(_capi.Outputs) == 8 * C.sizeof(C.c_void_p)
    assert C.sizeof(_capi.Labels) == 6 * C.sizeof(C.c_void_p)
    assert C.sizeof(_capi.StepResult) == 8 + 4 * 3 + 4 * 16 + 4  # + tail padding to 8
    assert _capi.Config.seed.offset % 8 == 0


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(alignnet3d.EngineError, match="no CPU fallback"):
        alignnet3d.Engine()


def test_config_marshalling():
    from alignnet3d.engine import make_c_config, default_model_config
    c = make_c_config(default_model_config())
    assert (c.num_points, c.num_bins, c.backbone) == (1024, 50, 0)
    assert list(c.emb_conv.w[: c.emb_conv.n]) == [64, 128, 1024]
    assert abs(c.s1_keep - 0.7) < 1e-7 and c.accept_inverted_angle == 1
    bad = default_model_config()
    bad["model"]["backbone"] = "voxel"
    with pytest.raises(AssertionError):
        make_c_config(bad)


def test_unbuilt_loss_variants_are_rejected():
    """config.py accepts training.loss.loss = 'p2p' and loss.options.soft_angle_classes (models/tp8.py:253-274,357-407); the engine
    only builds the 'separate' hard-class loss, so marshalling such a config must fail instead of training another loss silently."""
    from alignnet3d.engine import make_c_config, default_model_config
    cfg = default_model_config()
    cfg["training"]["loss"] = {"loss": "separate", "options": {"soft_angle_classes": False}}
    make_c_config(cfg)
    cfg["training"]["loss"]["loss"] = "p2p"
    with pytest.raises(AssertionError, match="separate"):
        make_c_config(cfg)
    cfg["training"]["loss"] = {"loss": "separate", "options": {"soft_angle_classes": True}}
    with pytest.raises(AssertionError, match="soft_angle_classes"):
        make_c_config(cfg)


def test_inference_kernels_are_spill_free():
    """The fused inference backbones issue their weight stream by hand (inline-asm global loads retired with counted waits,
    csrc/kernels_infer.h mfma_rows): a register the allocator spills or re-assigns while such a load is in flight is silently
    corrupted, so these kernels must stay spill-free -- and scratch written once per workgroup was 1.3 GB of HBM traffic per
    dgcnn_fused launch in round 1.  The build keeps hipcc's per-kernel resource remarks next to the objects (csrc/Makefile)."""
    import re
    path = os.path.join(ROOT, "alignnet-3d_amd", "csrc", "alignnet_api.remarks")
    if not os.path.exists(path):
        pytest.skip("no resource remarks next to the objects (library built by an older Makefile)")
    text = open(path).read()
    rows = re.findall(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?VGPRs Spill: (\d+)",
                      text, re.S)
    seen = {}
    for name, vgpr, scratch, occ, spill in rows:
        for key in ("pointnet_fused", "dgcnn_fused", "pointnet_split", "dgcnn_split", "fc_mfma", "knn_kernel"):
            if key in name:
                seen.setdefault(key, []).append((name, int(vgpr), int(scratch), int(occ), int(spill)))
    assert set(seen) == {"pointnet_fused", "dgcnn_fused", "pointnet_split", "dgcnn_split", "fc_mfma", "knn_kernel"}, sorted(seen)
    for key, lst in seen.items():
        for name, vgpr, scratch, occ, spill in lst:
            assert spill == 0 and scratch == 0, (name, vgpr, scratch, spill)
    # the shipped-shape DGCNN kernel runs three workgroups of eight waves per CU (52.5 KiB of LDS each): <= 80 registers per lane
    shipped = [r for r in seen["dgcnn_fused"] if "Li68ELi132E" in r[0]]
    assert shipped and shipped[0][1] <= 80 and shipped[0][3] >= 6, shipped
    # the kNN kernel for N <= 4096 keeps its 64 candidate distances per lane in registers at four waves per SIMD
    knn64 = [r for r in seen["knn_kernel"] if "ILi64E" in r[0]]
    assert knn64 and knn64[0][1] <= 128 and knn64[0][3] >= 4, knn64


def test_wide_phase3_kernels_are_spill_free():
    """train_fwd_phase3_wide carries the same hand-issued weight stream (mfma_rows<4, true, true>) and must not spill either; its bf16
    sibling keeps 8 weight fragments + 4 hidden-layer fragments + 2 Gram accumulators + 4 lift accumulators live and has no registers to
    lose (csrc/kernels_train_fwd_wide.h)."""
    import re
    path = os.path.join(ROOT, "alignnet-3d_amd", "csrc", "alignnet_train.remarks")
    if not os.path.exists(path):
        pytest.skip("no resource remarks next to the objects (library built by an older Makefile)")
    rows = re.findall(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?VGPRs Spill: (\d+)",
                      open(path).read(), re.S)
    wide = [(n, int(v), int(sc), int(o), int(sp)) for n, v, sc, o, sp in rows if "train_fwd_phase3_wide" in n]
    assert len(wide) == 5, wide
    for name, vgpr, scratch, occ, spill in wide:
        assert spill == 0 and scratch == 0 and occ >= 2, (name, vgpr, scratch, occ, spill)


def test_small_step_kernels_use_no_scratch():
    """The latency-bound launches of a training step (reductions, finishes, the heads' products, the loss) must not touch private memory: round 4
    found loss_prep_kernel indexing a six-element local array by a run-time variant index (64 bytes of scratch per lane, ~2 us), and a hoisted
    parameter load that made hipcc spill 394 registers in prep_hidden_reduce_kernel (10 -> 47 us) -- neither fails a numerical test."""
    import re
    path = os.path.join(ROOT, "alignnet-3d_amd", "csrc", "alignnet_train.remarks")
    if not os.path.exists(path):
        pytest.skip("no resource remarks next to the objects (library built by an older Makefile)")
    rows = re.findall(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+).*?VGPRs Spill: (\d+)", open(path).read(), re.S)
    keys = ("gemm_small", "gemm_tile64", "gemm_qimg", "bn_rows_fwd", "bn_rows_bwd", "stat_finish", "stat3_pool_finish", "stat2_from_gram", "reduce_multi",
            "prep3_kernel", "prep_hidden_reduce", "loss_prep", "loss_final", "dg_b0_totals", "dg_b0_cloud", "sparse_dw_jobs", "combine_dw_jobs",
            "centre_gram_jobs", "adam_kernel", "momentum_kernel", "pack_bf16_jobs", "pn_moments", "train_fwd_gram1")
    seen = set()
    for name, scratch, spill in rows:
        for k in keys:
            if k in name:
                seen.add(k)
                assert int(scratch) == 0 and int(spill) == 0, (name, scratch, spill)
    assert seen == set(keys), sorted(set(keys) - seen)


def test_big_training_passes_stay_within_their_scratch_budget():
    """The long passes of the training step in their SHIPPED instantiations (widths 64 / 128 compiled in): register file full (two waves per SIMD by LDS
    anyway), scratch zero or the two-to-seven spilled registers measured in round 6 -- a regression guard, since a change that pushes one of them into tens of
    spills (B2's run-time-width form sits at 78 - 81) costs 10 - 20 % of the pass and fails no numerical test.  VERDICT round 5, next 2 (i)."""
    import re
    path = os.path.join(ROOT, "alignnet-3d_amd", "csrc", "alignnet_train.remarks")
    if not os.path.exists(path):
        pytest.skip("no resource remarks next to the objects (library built by an older Makefile)")
    rows = re.findall(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?VGPRs Spill: (\d+)",
                      open(path).read(), re.S)
    budget = {   # mangled-name fragment -> (max scratch bytes / lane, max spilled VGPRs)
        "train_bwd_b2ILb0ELb1ELb0ELi64ELi128EE": (12, 2),     # pass B2, bf16, shipped widths (configs[2]'s longest kernel)
        "train_bwd_b2ILb0ELb0ELb0ELi64ELi128EE": (0, 0),      # pass B2, fp32
        "train_bwd_b2ILb0ELb0ELb1ELi64ELi128EE": (0, 0),      # pass B2 on given features (dgcnn point conv), fp32
        "train_bwd_b2ILb0ELb1ELb1ELi64ELi128EE": (0, 0),      # ... bf16
        "train_bwd_b1_bf16E": (0, 0),
        "train_bwd_b1ILi64ELi128ELb1EE": (0, 0),
        "dg_train_bwd_edgeILi64ELi128ELb0EE": (12, 2),         # configs[4]: 45 % of the fp32 dgcnn step
        "dg_train_bwd_edgeILi64ELi128ELb1EE": (0, 0),
        "dg_train_bwd_edge_denseILi64ELi128EE": (0, 0),
        "dg_train_fwdILi64ELb0EE": (36, 8),
        "dg_train_fwdILi64ELb1EE": (0, 0),
    }
    seen = set()
    for name, vgpr, scratch, occ, spill in rows:
        for frag, (smax, pmax) in budget.items():
            if frag in name:
                seen.add(frag)
                assert int(scratch) <= smax and int(spill) <= pmax and int(occ) >= 2, (name, vgpr, scratch, occ, spill)
    assert seen == set(budget), sorted(set(budget) - seen)


def test_no_register_is_rewritten_while_a_load_into_it_is_in_flight():
    """The hand-issued weight stream of csrc/kernels_infer.h (mfma_rows: asm `global_load_dwordx4` retired by counted asm waits) is invisible to
    the compiler's wait bookkeeping, so only the source's data flow keeps the allocator from giving a stream register to a new value before the
    wait.  Round 3 found `v_mov_b32 v142, 0` in front of the final drain while `global_load_dwordx4 v[142:145]` was outstanding (nothing spilled:
    the spill-free assertions above did not see it); the drain now names the stream registers.  tools/inflight_scan.py walks the disassembly of
    the built code objects: every vector load is followed through straight-line code up to the wait that retires it, and no instruction in
    between may write its destination."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import inflight_scan as S
    # the scanner itself, on the instruction sequence that was the bug and on its fixed form
    bug = "0000000000001000 <k>:\n\tglobal_load_dwordx4 v[142:145], v[186:187], off\n\ts_waitcnt vmcnt(2)\n\tv_mov_b32_e32 v142, 0\n\ts_waitcnt vmcnt(0)\n\ts_endpgm\n"
    ok = "0000000000001000 <k>:\n\tglobal_load_dwordx4 v[142:145], v[186:187], off\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v142, 0\n\ts_endpgm\n"
    assert len(S.scan(bug)[2]) == 1 and len(S.scan(ok)[2]) == 0
    objdump = os.path.join(S.LLVM, "llvm-objdump")
    objs = [os.path.join(ROOT, "alignnet-3d_amd", "csrc", n) for n in ("alignnet_api.o", "alignnet_train.o")]
    if not os.path.exists(objdump) or not all(os.path.exists(o) for o in objs):
        pytest.skip("no llvm-objdump / no objects next to the sources")
    for o in objs:
        nfun, nload, findings = S.scan(S.disassemble(o))
        assert nfun > 10 and nload > 100, (o, nfun, nload)
        assert not findings, findings[:5]


def test_shipped_library_reads_no_environment_and_carries_no_ablation_switch():
    """Round 3's library read ~20 ALIGNNET_* variables per training step, some of which (ALIGNNET_DBG) skipped work inside the kernels:
    wrong gradients at a faster time, one variable away from a benchmark.  The shipped library now imports no getenv at all (A/B
    kernel variants are alignnet_set_option "ab_*" keys, reported by bench.py) and the result-changing switches exist only in the
    separate ablation build (csrc/ablate.h, `make ablate`): the ablation macro must be the constant 0 in this one."""
    import subprocess
    lib = os.path.join(ROOT, "alignnet-3d_amd", "libalignnet_hip.so")
    nm = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True)
    assert nm.returncode == 0, nm.stderr
    undefined = {l.split()[-1].split("@")[0] for l in nm.stdout.splitlines() if l.strip()}
    assert "getenv" not in undefined and "secure_getenv" not in undefined, "the shipped library must not read the environment"
    raw = open(lib, "rb").read()
    for name in (b"ALIGNNET_DBG", b"ALIGNNET_P3_NOGRAM", b"ALIGNNET_NO_DEFER", b"ALIGNNET_B1_LEGACY", b"ablate_dbg"):
        assert name not in raw, name
    src = os.path.join(ROOT, "alignnet-3d_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".h", ".hip")) and f != "ablate.h":
            text = open(os.path.join(src, f)).read()
            if f.endswith(".h"):   # device code (the host files only derive the stamp pointers from h->ablate_dbg, 0 in this build)
                assert not re.search(r"\ba\.dbg\s*&", text), f"{f}: ablation test outside ALN_ABL()"
            for m in re.finditer(r"getenv\(", text):
                guard = text.rfind("#ifdef ALIGNNET_ABLATE", 0, m.start())
                assert guard >= 0 and text.find("#endif", guard, m.start()) < 0, f"{f}: getenv outside an ALIGNNET_ABLATE block"


def test_kernel_ids_and_option_keys_agree_between_header_engine_and_python():
    """`last_backbone_kernel` reports an ALIGNNET_KERNEL_* number (include/alignnet_hip.h); the Python side maps it to a name the -m gpu tests
    assert on (alignnet3d/engine.py KERNEL_IDS).  Every id of the header has exactly one name and vice versa, and every `ab_*` key the
    header's option list names is one csrc/engine.h dispatches on."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "alignnet-3d_amd"))
    from alignnet3d.engine import KERNEL_IDS
    hdr = open(os.path.join(ROOT, "include", "alignnet_hip.h")).read()
    ids = {name: int(num) for name, num in re.findall(r"#define ALIGNNET_KERNEL_(\w+) (\d+)", hdr)}
    assert sorted(ids.values()) == sorted(KERNEL_IDS.values()), (ids, KERNEL_IDS)
    assert len(set(KERNEL_IDS.values())) == len(KERNEL_IDS)
    eng = open(os.path.join(ROOT, "alignnet-3d_amd", "csrc", "engine.h")).read()
    keys = set(re.findall(r'\{"(ab_\w+)", AB_\w+\}', eng))
    named = set(re.findall(r'"(ab_\w+)"', hdr)) - {"ab_tiles_per_wg", "ab_mask"}
    assert named <= keys, named - keys
    assert "ab_split_tilewise" in keys


def test_profile_summaries_count_the_persistent_split_kernel_once():
    """tools/summarize_prof.py folds pointnet_split_persist<4> / <2> into one kernel name: bench.py's timers and `roofline.launches_per_step`
    (3 per step) count the backbone launches together, and tests/test_bench_launcher_cpu.py compares the two."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import importlib
    sp = importlib.import_module("summarize_prof")
    assert sp.short("void alignnet::pointnet_split_persist<4>(alignnet::SplitArgs)") == "void pointnet_split_persist"
    assert sp.short("void alignnet::pointnet_split_persist<2>(alignnet::SplitArgs)") == "void pointnet_split_persist"
    assert sp.short("void alignnet::pointnet_split<64, 128>(alignnet::SplitArgs)") == "void pointnet_split<64, 128>"
