"""The default encode kernels must not use scratch memory: register spills and device-function call frames
were half of the HBM traffic of the path until they were removed (profiles r01h vs r01i).
hipcc cross-compiles for gfx950 without a GPU, so this is checked on the CPU."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "slimfastq_amd", "csrc")
HOT = {"frame.hip": ("k_frame",), "models_w.hip": ("k_qlt_encode_k2", "k_rec_encode_w_fast"), "models_k.hip": ("k_gen_encode_kILi2E",),
       # the default path: one chain per lane over frozen tables (encoders and decoders)
       "chains.hip": ("k_qlt_encode_c", "k_gen_encode_c", "k_rec_encode_fILj62E", "k_rec_encode_fILj94E", "k_rec_encode_fILj127E",
                      "k_qlt_decode_c", "k_gen_decode_c", "k_rec_decode_f", "k_rec_tokens", "k_rec_code", "k_rec_dsym", "k_rec_dtext"),
       # the bases' match model (round 5)
       "gm.hip": ("k_gm_stage", "k_gm_insert", "k_gm_plan", "k_gm_price", "k_gm_code", "k_gm_decode_c")}


def kernel_metadata(src, tmp_path):
    out = tmp_path / (src + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S",
                    "-o", str(out), os.path.join(CSRC, src)], check=True, capture_output=True, timeout=900)
    meta = {}
    for blk in out.read_text().split("  - ."):
        name = re.search(r"\.name:\s+(\S+)", blk)
        vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
        sc = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
        if name and vg and sc:
            meta[name.group(1)] = (int(vg.group(1)), int(sc.group(1)))
    return meta


@pytest.mark.parametrize("src", sorted(HOT))
def test_default_kernels_use_no_scratch(src, tmp_path):
    meta = kernel_metadata(src, tmp_path)
    for want in HOT[src]:
        hits = [(k, v) for k, v in meta.items() if want in k]
        assert hits, "kernel %s not found in %s" % (want, src)
        for name, (vgprs, scratch) in hits:
            assert scratch == 0, "%s keeps %d bytes of scratch per lane (spills or call frames)" % (name, scratch)
            assert vgprs <= 128, "%s needs %d VGPRs: fewer than 4 waves per SIMD" % (name, vgprs)
