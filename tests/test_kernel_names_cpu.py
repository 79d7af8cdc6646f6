"""Every kernel the decoder can ask the code object for exists in it (no GPU needed: the names are read out of the built
metalchat.hsaco).  The host forms GEMV names from (format, dtype, arithmetic, linear-order row length, prologue, epilogue) in
decoder.cc::gemv(); a combination it may form but gemv_kernels.hip does not instantiate would only fail on the GPU, for the
one model shape that reaches it."""
import itertools
import os
import subprocess

import pytest

import metalchat_amd
from metalchat_amd import build as b

READELF = ["/opt/rocm/lib/llvm/bin/llvm-readelf", "/usr/bin/readelf"]


@pytest.fixture(scope="module")
def symbols():
    hsaco, _ = b.build_all()
    tool = next((t for t in READELF if os.path.exists(t)), None)
    if tool is None:
        pytest.skip("no readelf available")
    out = subprocess.check_output([tool, "--symbols", "--wide", hsaco], text=True)
    return {line.split()[-1] for line in out.splitlines() if " FUNC " in line}


# prologue / epilogue pairs run_layers(), run_head() and mc_decoder_time_gemv() launch (gemv_kernels.hip MC_GEMV_SET)
PE = ["p0_e0", "p1_e0", "p0_e1", "p1_e2", "p1_e3", "p1_e4", "p2_e0", "p2_e3"]
PE_FOLD = ["p3_e0", "p3_e1"]  # P.V range sums added by the Wo prologue: linear-order kernels only
PE_PICK = ["p1_e5", "p2_e5"]  # the greedy pick inside the output head's launch (gemv.h EPI_STORE_PICK): linear-order kernels only


def test_classic_gemv_family_is_complete(symbols):
    fams = ["i4_bfloat", "i4_bfloat_fast", "i4_bfloat_m4", "i4_bfloat_m4d", "i4_float", "i8_bfloat", "i8_float", "w_bfloat", "w_float"]
    missing = [f"mc_gemv_{f}_{pe}" for f, pe in itertools.product(fams, PE) if f"mc_gemv_{f}_{pe}" not in symbols]
    assert not missing, missing


def test_linear_order_families_are_complete(symbols):
    want = []
    for nch in (1, 2, 4, 7, 12, 14):  # decoder.cc lin_ok(): int4 rows of whole KiB
        want += [f"mc_gemv_i4_bfloat_lin{nch}_{pe}" for pe in PE + PE_FOLD + PE_PICK]
    for nch in (4, 14):  # decoder.cc ling_kib(): int8
        want += [f"mc_gemv_i8_bfloat_ling{nch}_{pe}" for pe in PE + PE_FOLD + PE_PICK if not pe.startswith("p2")]
    for nch in (4, 8, 11, 16):  # ... and plain bfloat weights
        want += [f"mc_gemv_w_bfloat_ling{nch}_{pe}" for pe in PE + PE_FOLD + PE_PICK if not pe.startswith("p2")]
    # rows of 1.5 KiB, two to a super row (gemv.h LSPLIT; decoder.cc lin_split_ok: K = 3072)
    want += [f"mc_gemv_i4_bfloat_lin3s_{pe}" for pe in PE]
    missing = [n for n in want if n not in symbols]
    assert not missing, missing


def test_decode_and_reference_kernels_present(symbols):
    for t in ("bfloat", "float"):
        for k in ("mc_attn_scores", "mc_attn_pv", "mc_attn_pv_reduce", "mc_embed", "mc_embed_q8", "mc_argmax", "mc_rope_kv",
                  "mc_rmsnorm_row", "mc_topk_candidates", "mc_sample"):
            assert f"{k}_{t}" in symbols, f"{k}_{t}"
    for k in ("mc_step_set", "mc_step_advance", "mc_step_rope", "mc_rope_table", "mc_argmax_keys",
              "mc_attn_fused_bfloat",  # decode attention in one launch
              # ... with the Wo GEMV behind it (decoder.cc attn_wo_fused: head_dim x KiB per Wo row)
              "mc_attn_wo_i4_bfloat_hd128_k2", "mc_attn_wo_i4_bfloat_hd64_k1", "mc_attn_wo_i4_bfloat_hd256_k2",
              "mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2",  # wq|wk|wv, attention and Wo in one launch (round 4)
              "mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2_t2", "mc_attn_qkv_wo_i4_bfloat_hd128_k2_q2_t4",  # ... 128- / 256-slot ranges (attn_qkv_wo_i4_wide_tiles)
              "mc_attn_qkv_wo_w_bfloat_hd64_k4_q4",    # ... for plain bfloat weights (decoder.cc attn_qkv_wo_w_fused)
              "mc_attn_qkv_wo_w13_w_bfloat_hd64_k4_q4_f3p3", "mc_attn_qkv_wo_w13_w_bfloat_hd64_k4_q4_f4p4",  # ... + ffn_norm + w1|w3 + act*mul (round 6: attn_qkv_wo_w13_w_fetch)
              "mc_gemv_i4_bfloat_lin12k4_p0_e0", "mc_gemv_i4_bfloat_lin12k4_p0_e1",  # Gemma-7B's w2: the K range of a pair over four waves (gemv_ksplit.h)
              "mc_attn_qkv_i4_bfloat_hd128_q4",        # ... without Wo, rows of 4 KiB (Llama-3-70B; decoder.cc attn_qkv_only_ok)
              "mc_attn_qkv_wo_i8_bfloat_hd128_k4_q4_t1", "mc_attn_qkv_wo_i8_bfloat_hd128_k4_q4_t2", "mc_attn_qkv_wo_i8_bfloat_hd128_k4_q4_t4",
              "mc_attn_qkv_wo_w_bfloat_hd64_k4_q4_t2", "mc_attn_qkv_wo_w_bfloat_hd64_k4_q4_t4",   # wide ranges of the plain-bfloat launch (attn_qkv_wo_w_tiles)  # ... for int8 weights (attn_qkv_wo_i8_tiles)
              "mc_attn_fused_qkn_bfloat",              # gemma3: q/k-norm + rope + cache write inside the one-launch attention
              "mc_attn_wo_qkn_i4_bfloat_hd256_k2_t1", "mc_attn_wo_qkn_i4_bfloat_hd256_k2_t2",  # ... with Wo in the launch too (attn_wo_qkn_tiles)
              "mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p1_t2", "mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p2_t2",
              "mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p1_t1", "mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p2_t1",
              "mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p1_t4", "mc_attn_qkv_wo_qkn_i4_bfloat_hd256_k2_p2_t4", "mc_attn_wo_qkn_i4_bfloat_hd256_k2_t4",  # ... and wq|wk|wv + the norms (attn_qkv_wo_qkn_ok)
              "mc_attn_fused_t2_bfloat",               # 128-slot ranges (decoder.cc attn_fused_t2: S = 8192)
              # the prompt pass on the quad-interleaved weight copy and its consumers with the split-K reduce inside (round 4)
              "mc_pf2_repack_i4", "mc_pf2_gemm_i4_bfloat", "mc_pf_rope_cache_parts_bfloat", "mc_pf_rope_cache_v4_bfloat", "mc_pf_rope_cache_parts_v4_bfloat", "mc_pf_act_mul_parts_bfloat",
              "mc_pf_rmsnorm_parts_bfloat", "mc_pf_rmsnorm2_parts_bfloat", "mc_pf_splitk_reduce_bfloat",
              "mc_exp_table_bfloat", "mc_pf_exp_window_bfloat", "mc_pf_attn2_bfloat_hd128", "mc_pf_attn4_bfloat_hd128", "mc_pf_attn8_bfloat_hd128", "mc_pf_attn8_bfloat_hd256", "mc_pf_attn8_bfloat_hd64_h8", "mc_pf_attn8_bfloat_hd64_h4",  # exp of a bfloat16 by table (prompt attention, silu)
              "mc_pf_dequant_rows_i4_bfloat", "mc_pf_dequant_rows_i8_bfloat"):  # the dequantised copy the opt-in library GEMM multiplies by
        assert k in symbols, k
    # the 256 x 256 ping-pong GEMM of long prompts (kernels/pf_gemm8.h, decoder.cc g8_launch)
    assert "mc_gelu_table_bfloat" in symbols
    for f in ("i4", "i8", "w"):
        for e in range(5):   # (e4, round 6: gemma's gelu(w1 x) * (w3 x) from the table)
            assert f"mc_pf_gemm8_{f}_bfloat_e{e}" in symbols
    # test aids live in a code object of their own (tests/kernels/): the product code object carries none
    assert not [k for k in symbols if k.startswith("mc_test_")], [k for k in symbols if k.startswith("mc_test_")]
    # the reference's own kernel names (ABI part 1, kernel/kernel.h:30-90)
    for k in ("rmsnorm_bfloat", "softmax_bfloat", "bmm_8_float", "hadamard_broadcast_bfloat_int8_t_float"):
        assert k in symbols, k


def test_no_hot_kernel_keeps_private_memory():
    """Round 4 found two hot kernels whose objects lived in SCRATCH without a single spill being reported: the q/k-norm policy of
    mc_attn_fused_qkn_bfloat (a run-time index into a small array of the policy object: 16.9 -> 13.4 us once it was gone) and the
    tiled prompt GEMM for plain bfloat weights (HIP uint4 structs that are only copied global -> register -> LDS are lowered as a
    memcpy through private memory: a 512-token prompt 4.58 -> 3.23 ms).  Neither shows in a timing until somebody looks; the code
    object's metadata says it for free: every kernel on the decode and prompt paths must report a private segment of zero bytes
    and no spills."""
    hsaco, _ = b.build_all()
    tool = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(tool):
        pytest.skip("no llvm-readelf")
    out = subprocess.check_output([tool, "--notes", hsaco], text=True)
    name, bad = None, []
    fields = {}
    for line in out.splitlines():
        line = line.strip()
        if line.startswith("- .") or line.startswith(".") or line.startswith("-"):
            key, _, val = line.lstrip("- ").partition(":")
            key, val = key.strip(), val.strip()
            if key == ".name":
                name = val
                fields[name] = {}
            elif name and key in (".private_segment_fixed_size", ".vgpr_spill_count", ".sgpr_spill_count"):
                fields[name][key] = int(val)
    hot = [n for n in fields if n.startswith(("mc_gemv_", "mc_attn_", "mc_pf", "mc_embed", "mc_argmax", "mc_topk_"))]
    assert len(hot) >= 250, len(hot)
    for n in hot:
        f = fields[n]
        if f.get(".private_segment_fixed_size", 0) or f.get(".vgpr_spill_count", 0) or f.get(".sgpr_spill_count", 0):
            bad.append((n, f))
    assert not bad, bad
