"""CPU: the C-ABI library loads and exports every symbol include/hpfg_hip.h declares; the ctypes structs match the header."""
import ctypes
import os
import re

from hpfg_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "hpfg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hpfg_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = L.load()
    names = _header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in L.PROTOTYPES, f"{n} has no ctypes prototype"
    assert sorted(L.PROTOTYPES) == names
    assert lib.hpfg_version() == L.VERSION


def test_struct_layouts_match_header_field_order():
    src = open(os.path.join(ROOT, "include", "hpfg_hip.h")).read()
    for cname, pyt in (("HpfgAct", L.Act), ("HpfgConvArgs", L.ConvArgs), ("HpfgWgradArgs", L.WgradArgs), ("HpfgPackDesc", L.PackDesc),
                       ("HpfgLossArgs", L.LossArgs), ("HpfgAugSample", L.AugSample), ("HpfgSlabDesc", L.SlabDesc),
                       ("HpfgPredBlocks", L.PredBlocks), ("HpfgFusedBwdArgs", L.FusedBwdArgs), ("HpfgPeerX", L.PeerX),
                       ("HpfgPeerBuf", L.PeerBuf)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(",")
            first = names[0].split()[-1].lstrip("*")
            first = re.sub(r"\[\w+\]$", "", first)          # array members: `const float* p[8]`, `void* mbox[HPFG_PEER_MAX_RANKS]`
            fields.append(first)
            fields += [n.strip().lstrip("*") for n in names[1:]]
        assert fields == [f[0] for f in pyt._fields_], (cname, fields, [f[0] for f in pyt._fields_])


def test_argument_errors_are_reported_not_crashed():
    lib = L.load()
    assert lib.hpfg_conv_fwd(None, None) == -1
    assert b"null" in lib.hpfg_last_error()
    assert lib.hpfg_wgrad(None, None) == -1
    assert lib.hpfg_fused_bwd(None, None) == -1 and lib.hpfg_fused_bwd_grid(None) == -1
    assert lib.hpfg_peer_allreduce_f32(None, None, None) == -1 and b"null" in lib.hpfg_last_error()
    pb = L.PeerBuf()
    pb.world, pb.rank, pb.n, pb.slice = 9, 0, 16, 4
    assert lib.hpfg_peer_allreduce_f32(ctypes.byref(pb), ctypes.c_void_p(256), None) == -1          # more ranks than a node has
    pb.world, pb.n, pb.slice = 1, 16, 16
    assert lib.hpfg_peer_allreduce_f32(ctypes.byref(pb), ctypes.c_void_p(256), None) == 0           # one rank: nothing to exchange, nothing launched
    assert lib.hpfg_conv_stat_blocks(2, 32, 32) == 2 * 4 and lib.hpfg_conv_stat_blocks(1, 24, 24) == 12
    assert lib.hpfg_wgrad_splits(16, 224, 224, 16, 16, 9) >= 1
    assert ctypes.sizeof(L.Act) % 8 == 0


def test_host_side_size_queries():
    """Size / split queries are plain host arithmetic (no GPU call): the values the Python side sizes its workspaces with."""
    lib = L.load()
    # K-split of the split-bf16 GEMM: only products with few 128 x 128 output tiles and a long contraction (the 1568-token SegFormer layers)
    assert lib.hpfg_gemm_bf16x3_splits(1568, 32, 2048) == 16          # 13 tiles, K = 2048: capped at 16 splits of >= 128
    assert lib.hpfg_gemm_bf16x3_splits(1568, 256, 1024) == 8          # 26 tiles -> 256 / 26 = 9, K / 128 = 8
    assert lib.hpfg_gemm_bf16x3_splits(1568, 256, 256) == 2
    assert lib.hpfg_gemm_bf16x3_splits(100352, 256, 256) == 1         # 1568 tiles: enough workgroups
    assert lib.hpfg_gemm_bf16x3_splits(1568, 256, 64) == 1            # nothing to split
    # one BatchNorm partial-sum row per workgroup of the first layer, at most 1024
    assert lib.hpfg_conv_first_rows(16, 224, 224) == 1024 and lib.hpfg_conv_first_rows(2, 32, 32) == 8
    assert lib.hpfg_upsample2x_bwd_blocks(16, 112, 112, 16) % 8 == 0   # (the XCD-aware ranges need a multiple of 8 workgroups)
    # peer-window gradient exchange: one slice per rank, a multiple of 4 floats; window = 256 bytes of flags + inbox + result
    assert lib.hpfg_peer_buf_slice(8, 1001) == 128 and lib.hpfg_peer_buf_slice(2, 10) == 8 and lib.hpfg_peer_buf_slice(1, 7) == 8
    assert lib.hpfg_peer_buf_bytes(2, 10) == 256 + 2 * 2 * 8 * 4
    assert lib.hpfg_peer_slot_bytes(2, 512) > 0
