"""Every program a pass of host/passes.cpp names must be in the host layer's program table (a registration lost in an edit of
gpu.cpp — round 4 dropped "sssr_trace_windowed" that way — otherwise only shows on the GPU, as "Program not found" in the
first frame that takes the path)."""
import os
import re

from vk_renderer_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_program_the_passes_name_is_registered():
    src = open(os.path.join(ROOT, "vk-renderer_amd", "host", "passes.cpp")).read()
    names = sorted(set(re.findall(r'(?:create_compute_pipeline|set_program)\("([a-z_0-9]+)"\)', src)))
    assert len(names) >= 20, names
    lib = host.lib()
    missing = [n for n in names if not lib.vkrh_has_program(n.encode())]
    assert not missing, f"programs named by host/passes.cpp but unknown to host/gpu/gpu.cpp: {missing}"
    assert not lib.vkrh_has_program(b"no_such_program")
