"""Camera / projection conventions of the reference and the frozen synthetic frame setup.

Reference: scene/camera.hpp:14-39 (yaw/pitch camera, world-up (0,-1,0), glm::lookAt),
main.cpp:293-294 (eye (0,1,-1), glm::perspective(60 deg, W/H, 0.05, 80), RH, depth 0..1
because of GLM_FORCE_DEPTH_ZERO_TO_ONE, camera.hpp:4-5), main.cpp:368-373 (normal_mat).
glm is not vendored in the reference, so these matrices are restated from glm's published
formulas, computed in float64 and rounded to float32: the kernels take them as data.
"""
import ctypes as C
import math

import numpy as np

from . import abi

FOVY = math.radians(60.0)
ZNEAR, ZFAR = 0.05, 80.0
SEED = 0x5EED0001


def perspective_rh_zo(fovy, aspect, znear, zfar):
    t = math.tan(fovy / 2.0)
    m = np.zeros((4, 4), dtype=np.float64)
    m[0, 0] = 1.0 / (aspect * t)
    m[1, 1] = 1.0 / t
    m[2, 2] = zfar / (znear - zfar)
    m[3, 2] = -1.0
    m[2, 3] = -(zfar * znear) / (zfar - znear)
    return m


def look_at_rh(eye, center, up):
    eye, center, up = (np.asarray(v, dtype=np.float64) for v in (eye, center, up))
    f = center - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4, dtype=np.float64)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -s.dot(eye), -u.dot(eye), f.dot(eye)
    return m


def camera_view(pos, yaw_deg=90.0, pitch_deg=0.0, world_up=(0.0, -1.0, 0.0)):
    """scene/camera.hpp:26-39"""
    yaw, pitch = math.radians(yaw_deg), math.radians(pitch_deg)
    f = np.array([math.cos(yaw) * math.cos(pitch), math.sin(pitch), math.sin(yaw) * math.cos(pitch)])
    f /= np.linalg.norm(f)
    right = np.cross(f, np.asarray(world_up, dtype=np.float64))
    right /= np.linalg.norm(right)
    up = np.cross(right, f)
    up /= np.linalg.norm(up)
    pos = np.asarray(pos, dtype=np.float64)
    return look_at_rh(pos, pos + f, up)


class FrameSetup:
    """All per-frame uniforms of the frozen benchmark frame: current camera at
    (0,1,-1) yaw 90; previous frame = eye + (0.02,0,0.01), yaw + 0.2 deg; jitter 0;
    angle_offset = 60/360, weight_ratio 1, max_roughness 1, render_flags 7, accumulate 1."""

    def __init__(self, width, height, frame_random=0, use_mis=1, eye=(0.0, 1.0, -1.0), yaw=90.0,
                 prev_delta=(0.02, 0.0, 0.01), prev_yaw_delta=0.2, material="flat"):
        self.width, self.height = width, height
        # "flat": one roughness per object (the frozen scene of SURVEY 8(d)); "textured": per-texel roughness
        # (VKR_SYNTH_TEXTURED_ROUGHNESS, include/vkr_postfx.h)
        if material not in ("flat", "textured"):
            raise ValueError(f"material {material!r}: 'flat' or 'textured'")
        self.material = material
        self.synth_flags = abi.SYNTH_TEXTURED_ROUGHNESS if material == "textured" else 0
        self.aspect = float(width) / float(height)
        # glm computes in float32; here every *input* matrix is rounded to float32 first and every
        # derived matrix is evaluated in float64 from those and rounded once (host/glm_compat.hpp does
        # the same), so the Python and the C++ host layers hand bit-identical uniforms to the kernels.
        f32 = lambda m: np.asarray(m, dtype=np.float32).astype(np.float64)
        self.proj = f32(perspective_rh_zo(FOVY, self.aspect, ZNEAR, ZFAR))
        self.view = f32(camera_view(eye, yaw))
        peye = tuple(e + d for e, d in zip(eye, prev_delta))
        self.prev_view = f32(camera_view(peye, yaw + prev_yaw_delta))
        self.mvp = f32(self.proj @ self.view)
        self.prev_mvp = f32(self.proj @ self.prev_view)
        self.inv_view = f32(np.linalg.inv(self.view))
        self.prev_inv_view = f32(np.linalg.inv(self.prev_view))
        self.normal_mat = self.inv_view.T  # transpose(inverse(view)), main.cpp:368
        self.prev_normal_mat = self.prev_inv_view.T
        self.frame_random = frame_random
        self.use_mis = use_mis
        self.fazz = (np.float32(FOVY), np.float32(self.aspect), np.float32(ZNEAR), np.float32(ZFAR))

    def _fazz4(self):
        return (C.c_float * 4)(*[float(v) for v in self.fazz])

    def synth(self, prev=False, depth_only=False):
        p = abi.SynthParams()
        if prev:  # the previous frame's own G-buffer (only its depth is consumed)
            p.camera_to_world = abi.Mat4.from_np(self.prev_inv_view)
            p.mvp = abi.Mat4.from_np(self.prev_mvp)
            p.prev_mvp = abi.Mat4.from_np(self.prev_mvp)
        else:
            p.camera_to_world = abi.Mat4.from_np(self.inv_view)
            p.mvp = abi.Mat4.from_np(self.mvp)
            p.prev_mvp = abi.Mat4.from_np(self.prev_mvp)
        p.fovy, p.aspect, p.znear, p.zfar = [float(v) for v in self.fazz]
        p.seed = SEED
        p.flags = abi.SYNTH_DEPTH_ONLY if depth_only else self.synth_flags
        return p

    def gtao_params(self):
        p = abi.GtaoParams()
        p.normal_mat = abi.Mat4.from_np(self.normal_mat)
        p.fovy, p.aspect, p.znear, p.zfar = [float(v) for v in self.fazz]
        return p

    def gtao_push(self, angle_offset=60.0 / 360.0, weight_ratio=1.0, two_directions=0, reflections_only=0):
        return abi.GtaoPush(angle_offset, weight_ratio, self.use_mis, two_directions, reflections_only)

    def gtao_filter_push(self):
        return abi.GtaoFilterPush(float(self.fazz[2]), float(self.fazz[3]))

    def gtao_accum_params(self):
        p = abi.GtaoAccumParams()
        p.inverse_camera = abi.Mat4.from_np(self.inv_view)
        p.prev_inverse_camera = abi.Mat4.from_np(self.prev_inv_view)
        p.mvp = abi.Mat4.from_np(self.mvp)
        p.fovy_aspect_znear_zfar = self._fazz4()
        return p

    def trace_params(self, frame_random=None):
        p = abi.TraceParams()
        p.normal_mat = abi.Mat4.from_np(self.normal_mat)
        p.frame_random = self.frame_random if frame_random is None else frame_random
        p.fovy, p.aspect, p.znear, p.zfar = [float(v) for v in self.fazz]
        return p

    def reproject_params(self):
        p = abi.ReprojectParams()
        p.inverse_camera = abi.Mat4.from_np(self.inv_view)
        p.prev_inverse_camera = abi.Mat4.from_np(self.prev_inv_view)
        p.fovy_aspect_znear_zfar = self._fazz4()
        return p
