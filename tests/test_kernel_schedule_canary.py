"""A tripwire, not a parity test: the register count hipcc gives the headline backward kernel.

Round 5 (docs/kernels/K2_render_backward.md, "A kernel-argument layout regression"): one more integer in `RenderArgs`, in front of the other
fields, changed NOTHING in the source of `render_bwd_pair_k` -- and hipcc compiled it to a different schedule (73 instead of 82 VGPRs, fewer
tap loads in flight) that measured 13.4-14.2 against 12.3-12.5 ms at cfg3.  The compiler's schedule of this kernel is sensitive to inputs
that have nothing to do with it, so a change of its register count is the cheapest signal that the schedule moved: when this test fails,
A/B the build against the previous one (profiles/r05d_ab_lib.sh) before updating the number."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "videoloop3d_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="no hipcc")
def test_headline_backward_keeps_its_schedule():
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
             "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only"]
    out = subprocess.run([HIPCC] + flags + ["-c", os.path.join(CSRC, "vl3d_render_c3_mpv_sig.hip"), "-o", os.devnull],
                         capture_output=True, text=True).stderr
    # the plain fp32 frame-pair backward of the shipped planar convention: render_bwd_pair_k<1,1,1,1,1,false,false,false,32>
    name = "_ZN12_GLOBAL__N_117render_bwd_pair_kILi1ELi1ELi1ELi1ELi1ELb0ELb0ELb0ELi32EEEvN18vl3d_render_detail10RenderArgsE"
    blocks = re.split(r"remark: Function Name: ", out)
    mine = [b for b in blocks if b.startswith(name)]
    assert mine, "the frame-pair backward is no longer instantiated under this name (update the tripwire)"
    vgprs = int(re.search(r"VGPRs: (\d+)", mine[0]).group(1))
    scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", mine[0]).group(1))
    assert scratch == 0
    assert vgprs == 82, (f"render_bwd_pair_k now takes {vgprs} VGPRs (82 when it measured 11.7-12.5 ms at cfg3): its schedule moved -- "
                         "A/B this build against the previous one (profiles/r05d_ab_lib.sh) before accepting the new number")
