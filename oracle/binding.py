"""ORACLE — TEST INFRASTRUCTURE ONLY.  Parity: UNPINNED (see glsl.hpp).

Python binding of oracle/libvkr_oracle.so (the CPU restatement of the reference shaders) and the glue
that lets the test drivers run on it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
import this module; the product package (vk-renderer_amd/) contains no reference to it — its flat chain
driver and its tiling driver take a *backend* that this module registers / provides on request.
"""
import ctypes as C
import os
import subprocess

import torch

import vk_renderer_amd  # noqa: F401  (import shim for the hyphenated package directory)
from vk_renderer_amd import abi, chain, tiling
from vk_renderer_amd.images import mip_extent

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(HERE, "libvkr_oracle.so")
_lib = None


def load(build_if_missing=False):
    """The oracle library with the vkr_ref_* twins of every C-ABI entry typed (same arguments minus the stream)."""
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            if not build_if_missing:
                raise RuntimeError(f"{ORACLE_LIB} is missing: run `make -C oracle` (or __graft_entry__.build())")
            subprocess.check_call(["make", "-C", HERE, "-j8"])
        lib = C.CDLL(ORACLE_LIB)
        for name, args in abi.ENTRY_ARGS.items():
            if name in abi.SCHEDULE_VARIANTS:  # the same images as another entry by another launch schedule: nothing to restate
                continue
            fn = getattr(lib, "vkr_ref_" + name)
            fn.argtypes = args
            fn.restype = C.c_int
        lib.vkr_ref_halton23.argtypes = [C.c_void_p, C.c_uint32]
        lib.vkr_ref_halton23.restype = None
        lib.vkr_ref_threads.restype = C.c_int
        lib.vkr_ref_numeric_contract.restype = C.c_uint32
        lib.vkr_ref_set_threads.argtypes = [C.c_int]
        _lib = lib
    return _lib


def install(build_if_missing=False):
    """Registers the "oracle" backend with the flat chain driver: PostFxChain(..., backend="oracle") then runs
    every pass through vkr_ref_* on host memory."""
    lib = load(build_if_missing)
    chain.register_backend("oracle", lambda: (lib, "vkr_ref_", False))
    return lib


class OracleBackend:
    """The multi-GPU tiling driver's compute backend over the oracle (host memory, gloo in the CPU tests)."""

    def __init__(self, setup, window, tiled, device=None):
        install()
        self.chain = chain.PostFxChain(setup.width, setup.height, backend="oracle", setup=setup, window=window, force_tiled=tiled)
        self.frame = None
        self.device = torch.device("cpu")
        self.gather_mips = tiling.GATHER_MIPS

    def set_gather_mips(self, n):
        self.gather_mips = n

    # hit colours / hit normals by request / reply (tiling.TiledFrame gather_mode 0 / 2): the flat chain's twins of the C-ABI
    windowed = False  # set by the tiling driver: the trace leaves rays that end on another rank's rows pending

    def local_rows(self, normals):
        """own window rows into the whole-frame images (what the all-gather would have put there)"""
        c = self.chain
        for src, dst in ((c.albedo, c.frame_albedo),) + (((c.dn, c.frame_normals),) if normals else ()):
            assert src.pitch[0] == dst.pitch[0]
            o = src.origin[1] * dst.pitch[0]
            dst.host[o: o + src.height * src.pitch[0]] = src.host[: src.height * src.pitch[0]]

    def hit_count(self, bounds, normals):
        return self.chain.hit_count(bounds, normals)

    def hit_write(self, bounds, counts, normals):
        return self.chain.hit_write(bounds, counts, normals)

    def hit_reply(self, requests, count, normals):
        return self.chain.hit_reply(requests, count, normals)

    def hit_scatter(self, requests, replies, count, normals):
        self.chain.hit_scatter(requests, replies, count, normals)
        if normals:
            self.chain.ssr_validate()

    def rows(self, name, mip=0):
        img = getattr(self.chain, name)
        h, w = mip_extent(img.height, mip), mip_extent(img.width, mip)
        t = torch.from_numpy(img.host)[img.offset[mip]: img.offset[mip] + h * img.pitch[mip]].view(h, img.pitch[mip])
        return t, img.bpp, (img.origin[0] >> mip, img.origin[1] >> mip, w, h)

    def prepare(self):
        c = self.chain
        c.synth()
        c.build_prev_hiz()
        c.init_histories()
        c.preintegrate_pdf()

    def run_stage(self, stage):
        c = self.chain
        if stage == "downsample":
            c.downsample()
        elif stage == "taa":
            c.taa()
        elif stage == "trace":
            if c.tiled:
                c.hiz_tail(self.gather_mips)
            if self.windowed:
                c.ssr_trace_windowed(frame_random=c.frame_index % 16)
            else:
                c.ssr_trace(frame_random=c.frame_index % 16)
        elif stage == "gtao":
            c.gtao_main()
            c.gtao_filter()
            c.gtao_accumulate()
        elif stage == "ssr_resolve":
            c.ssr_filter()
            c.ssr_blur()
            c.frame_index += 1
        else:
            raise ValueError(stage)

    def run_all(self):
        self.chain.frame()

    def end_frame(self):
        self.chain.swap_histories()

    def sync(self):
        pass
