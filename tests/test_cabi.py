"""The C-ABI library loads on a CPU-only host and exports every symbol include/pvrl.h declares (no compute calls)."""
import ctypes
import os

from procedurevrl_amd import _lib


def test_header_parses_and_lists_entry_points():
    protos = _lib.parse_header()
    assert len(protos) >= 24
    for must in ("pvrl_gemm_nt_bf16", "pvrl_gemm_tn_bf16", "pvrl_layernorm_fwd", "pvrl_layernorm_bwd", "pvrl_attn_fwd",
                 "pvrl_attn_bwd", "pvrl_attn_t8_fwd", "pvrl_attn_t8_bwd", "pvrl_patchify", "pvrl_kl_topk", "pvrl_mse",
                 "pvrl_adam_step", "pvrl_sgd_step"):
        assert must in protos
    for name, (ret, args) in protos.items():
        assert ret in ("int", "int64_t")
        for ty, _ in args:
            assert ty in _lib._CTYPES, (name, ty)   # plain pointers and sizes only: no torch types at the boundary


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from procedurevrl_amd.csrc import build_ext
        build_ext.build(verbose=False)
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.parse_header():
        assert hasattr(cdll, name), f"{name} declared in include/pvrl.h but not exported"
    L = _lib.lib()
    assert L.PVRL_EPI_RESID_F32 == 3 and L.PVRL_EPI_DQGELU == 6 and L.PVRL_EPI_RESID_16 == 7


def test_pure_size_queries_run_without_a_gpu():
    L = _lib.lib()
    assert L.call("pvrl_gemm_tn_workspace_bytes", 768, 768, 8) == 8 * (768 * 768 + 768) * 4 + 256
    # one workgroup per 4 rows + one (a split matrix, pvrl_rows, always has a workgroup for each of its parts), at most 512
    assert L.call("pvrl_layernorm_bwd_workspace_bytes", 100, 768) == 26 * 3 * 768 * 4
    assert L.call("pvrl_layernorm_bwd_workspace_bytes", 50208, 768) == 512 * 3 * 768 * 4


def test_grouped_weight_gradient_plan_runs_without_a_gpu():
    """pvrl_gemm_tn_grouped_plan_splits / _workspace_bytes are pure host functions: a transformer block's seven weight
    gradients (153 tiles of 256x256) are cut into 5 row slices = 765 work items = 2.99 rounds of 256 CUs; shapes that are
    not multiples of 128, an empty list and more than 8 problems are refused."""
    import ctypes as C
    L = _lib.lib()
    M = 50208
    dims = [(768, 3072), (3072, 768), (768, 768), (2304, 768), (768, 768), (768, 768), (2304, 768)]
    arr = (_lib.TnProblem * len(dims))()
    for a, (N, K) in zip(arr, dims):
        a.P, a.ldp, a.Q, a.ldq, a.M, a.N, a.K, a.beta, a.dW, a.dbias = 16, N, 16, K, M, N, K, 0.0, 16, None
    ap = C.addressof(arr)
    assert L.call("pvrl_gemm_tn_grouped_plan_splits", len(dims), ap) == 5
    want = sum(5 * (N * K + N) * 4 for N, K in dims)
    assert L.call("pvrl_gemm_tn_grouped_workspace_bytes", len(dims), ap, 5) == want
    arr[2].N = 640                                            # half tiles (128 mod 256) are staged with zero columns
    assert L.call("pvrl_gemm_tn_grouped_plan_splits", len(dims), ap) >= 1
    arr[2].N = 600                                            # not a multiple of 128
    assert L.call("pvrl_gemm_tn_grouped_plan_splits", len(dims), ap) == -1
    assert L.call("pvrl_gemm_tn_grouped_plan_splits", 0, ap) == -1
    assert L.call("pvrl_gemm_tn_grouped_plan_splits", 9, ap) == -1
    assert L.call("pvrl_mvit_pool_bwd_workspace_bytes") == (2048 * 27 * 96 + 2048 * 2 * 96) * 4


def test_product_path_fails_loudly_without_gpu():
    import pytest
    import torch
    from procedurevrl_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.PvrlError):
        ops.layernorm_fwd(torch.zeros(4, 768), torch.ones(768), torch.zeros(768), 1e-6)


def test_rel_operand_form_round_trips_on_the_host():
    """ops_mvit.rel_pack / rel_unpack (the hi | lo 16-bit pair pvrl_mvit_rel_fwd writes and pvrl_mvit_attn_* read): width from
    the C side, padding columns zero, value recovered to ~2^-16 -- pure host code plus two size queries."""
    import torch
    from procedurevrl_amd import ops_mvit as om
    assert om.rel_width((8, 7, 7)) == 32 and om.rel_width((8, 14, 14)) == 64
    L = _lib.lib()
    assert L.call("pvrl_mvit_attn_keymap_bytes", 8, 7, 7) == 13 * 4096 and L.call("pvrl_mvit_attn_keymap_bytes", 8, 14, 14) == 50 * 4096
    g = torch.Generator().manual_seed(0)
    for k_thw in ((8, 7, 7), (8, 14, 14)):
        J = sum(k_thw)
        rel = torch.randn(3, 10, J, generator=g) * 3
        for osc in (1.0, 96 ** 0.5):
            relp = om.rel_pack(rel, k_thw, osc)
            JP = om.rel_width(k_thw)
            assert relp.shape == (3, 10, 2 * JP)
            assert float(relp[..., J:JP].float().abs().max()) == 0.0 and float(relp[..., JP + J:].float().abs().max()) == 0.0
            back = om.rel_unpack(relp, k_thw, osc)
            assert float((back - rel).abs().max() / rel.abs().max()) < 2e-5
