"""Host-side mirror of rt::RayTracer (RayTracer/RayTracer.h:14-41) over the C ABI
(include/rt_mi355x.h).  Same method names, argument order, units and error behaviour as
the reference class, so that the parity tests read like a caller of the reference
(OpenGLView/MainFrame.cpp:45,219-256).

There is NO fallback: if librt_mi355x.so is missing or no HIP device is usable this
module raises.  PyTorch is not involved here at all.
"""
import ctypes as C
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_MI355X_LIB selects another build of the same library (kernel A/B experiments only)
_LIB_PATH = os.environ.get("RT_MI355X_LIB") or os.path.join(_HERE, "lib", "librt_mi355x.so")

MATH_FMA, MATH_STRICT = 0, 1
FLAG_NO_FILTER = 1
FLAG_NO_BINNING = 2
FLAG_NEAREST_HIT = 4
FLAG_SMOOTH_NORMALS = 8
FLAG_NO_MACRO_BINS = 16
FLAG_NO_SUPER_BINS = 64
FLAG_NO_SURE_HIT = 32
BUF_RENDER, BUF_COUNTS, BUF_IMAGE, BUF_RNG, BUF_FRAME = 0, 1, 2, 3, 4
GROUP_ID_BYTES = 128
TRANSPORT_RCCL, TRANSPORT_PEER = 0, 1

# every symbol include/rt_mi355x.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "rt_tracer_create", "rt_tracer_create_ex", "rt_tracer_destroy", "rt_tracer_trace",
    "rt_tracer_stop", "rt_tracer_resize", "rt_tracer_set_camera_parameters",
    "rt_tracer_rotate_camera", "rt_tracer_upload_scene", "rt_tracer_set_update_callback",
    "rt_tracer_set_finished_callback", "rt_tracer_wait", "rt_tracer_set_seed",
    "rt_tracer_upload_spheres", "rt_tracer_trace_enqueue", "rt_tracer_trace_enqueue_n", "rt_tracer_sync", "rt_tracer_trace_stats", "rt_tracer_launch", "rt_tracer_launch_iterations", "rt_tracer_fused_iterations", "rt_tracer_set_image_mirror", "rt_tracer_set_list_reuse", "rt_tracer_stream_b", "rt_tracer_upload_scene_edges", "rt_pack_normal",
    "rt_unpack_normal",
    "rt_tracer_kernel_time", "rt_tracer_launch_time", "rt_tracer_read_buffer", "rt_tracer_copy_buffer_to_device", "rt_tracer_copy_buffer_to_device_async",
    "rt_tracer_stream",
    "rt_tracer_device_pointer", "rt_tracer_buffer_bytes", "rt_tracer_info",
    "rt_tracer_last_error", "rt_last_error", "rt_device_count", "rt_version",
    "rt_dbg_hit_triangle", "rt_dbg_sincos", "rt_dbg_valu_peak", "rt_dbg_check_midrange", "rt_dbg_trace_occupancy", "rt_dbg_uniform", "rt_dbg_get_ray",
    "rt_dbg_rng_init_host",
    "rt_tracer_create_multi", "rt_group_unique_id", "rt_tracer_join_group", "rt_tracer_leave_group",
    "rt_tracer_gather_time", "rt_tracer_band_count", "rt_tracer_band_info",
    "rt_tracer_join_group_bands", "rt_balance_rows", "rt_tracer_rebalance", "rt_tracer_set_band", "rt_dbg_read_tile_lists", "rt_dbg_wave_list_counts", "rt_dbg_focal_boxes", "rt_dbg_classify",
    "rt_tracer_gather_only", "rt_tracer_group_info",
]


class RtError(RuntimeError):
    pass


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("full_height", C.c_uint32),
                ("row_begin", C.c_uint32), ("use_time_seed", C.c_uint32), ("math_mode", C.c_uint32),
                ("seed", C.c_uint64), ("flags", C.c_uint32), ("samples_in_flight", C.c_uint32),
                ("lds_chunk", C.c_uint32), ("bin_list", C.c_uint32), ("transport", C.c_uint32)]


CALLBACK = C.CFUNCTYPE(None, C.POINTER(C.c_uint32), C.c_size_t, C.c_void_p)

_lib = None
_lib_lock = threading.Lock()


def library_path():
    return _LIB_PATH


def _share_hip_runtime_with_torch():
    """One process must not hold two HIP runtimes: PyTorch wheels bundle their own
    libamdhip64.so (SONAME libamdhip64.so.7, the same as /opt/rocm's), and whichever copy
    initialises second finds no device.  The multi-GPU driver (dist.py) needs torch.distributed
    (RCCL) next to this library, so when a PyTorch install is present its libamdhip64 is mapped
    first -- WITHOUT importing torch -- and librt_mi355x.so's NEEDED libamdhip64.so.7 binds to
    that same object; a later `import torch` reuses it too.  RT_MI355X_HIP_RUNTIME=system
    keeps /opt/rocm's runtime (then do not use torch.cuda in the same process)."""
    if os.environ.get("RT_MI355X_HIP_RUNTIME", "") == "system":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return None
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            return cand
    except (ImportError, OSError, ValueError):
        pass
    return None


HIP_RUNTIME = None      # path of the HIP runtime shared with PyTorch, or None for /opt/rocm's


def load_library():
    """dlopen librt_mi355x.so and declare the ABI.  Raises if the extension is missing."""
    global _lib, HIP_RUNTIME
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise RtError("HIP extension not built: %s is missing (run `python -m raytracertest_amd.build` "
                          "or __graft_entry__.build()); there is no CPU fallback" % _LIB_PATH)
        HIP_RUNTIME = _share_hip_runtime_with_torch()
        L = C.CDLL(_LIB_PATH)
        vp, u32p, f32p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float)
        L.rt_tracer_create.argtypes = [u32p, f32p, f32p, C.c_float, C.c_float, C.c_float, C.POINTER(vp)]
        L.rt_tracer_create_ex.argtypes = [u32p, f32p, f32p, C.c_float, C.c_float, C.c_float,
                                          C.POINTER(Options), C.POINTER(vp)]
        L.rt_tracer_destroy.argtypes = [vp]
        L.rt_tracer_destroy.restype = None
        L.rt_tracer_trace.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
        L.rt_tracer_stop.argtypes = [vp]
        L.rt_tracer_stop.restype = None
        L.rt_tracer_resize.argtypes = [vp, u32p]
        L.rt_tracer_set_camera_parameters.argtypes = [vp, C.c_float, C.c_float, C.c_float]
        L.rt_tracer_set_camera_parameters.restype = None
        L.rt_tracer_rotate_camera.argtypes = [vp, f32p]
        L.rt_tracer_rotate_camera.restype = None
        L.rt_tracer_upload_scene.argtypes = [vp, vp, C.c_size_t]
        L.rt_tracer_upload_scene_edges.argtypes = [vp, vp, C.c_size_t]
        L.rt_pack_normal.argtypes = [f32p]
        L.rt_pack_normal.restype = C.c_float
        L.rt_unpack_normal.argtypes = [C.c_float, f32p]
        L.rt_unpack_normal.restype = None
        L.rt_tracer_set_update_callback.argtypes = [vp, CALLBACK, vp]
        L.rt_tracer_set_update_callback.restype = None
        L.rt_tracer_set_finished_callback.argtypes = [vp, CALLBACK, vp]
        L.rt_tracer_set_finished_callback.restype = None
        L.rt_tracer_wait.argtypes = [vp]
        L.rt_tracer_set_seed.argtypes = [vp, C.c_uint64]
        L.rt_tracer_upload_spheres.argtypes = [vp, vp, C.c_size_t]
        L.rt_tracer_trace_enqueue.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.rt_tracer_trace_enqueue_n.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
        L.rt_tracer_sync.argtypes = [vp]
        L.rt_tracer_launch.argtypes = [vp, C.c_uint32, C.c_int, C.c_int]
        L.rt_tracer_launch_iterations.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
        L.rt_tracer_fused_iterations.argtypes = [vp, C.c_uint32]
        L.rt_tracer_set_image_mirror.argtypes = [vp, vp]
        L.rt_tracer_set_list_reuse.argtypes = [vp, C.c_int]
        L.rt_tracer_trace_stats.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64)]
        L.rt_tracer_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
        L.rt_tracer_launch_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
        L.rt_tracer_read_buffer.argtypes = [vp, C.c_int, vp, C.c_size_t]
        L.rt_tracer_copy_buffer_to_device.argtypes = [vp, C.c_int, vp, C.c_size_t]
        L.rt_tracer_copy_buffer_to_device_async.argtypes = [vp, C.c_int, vp, C.c_size_t]
        L.rt_tracer_stream.argtypes = [vp]
        L.rt_tracer_stream.restype = vp
        L.rt_tracer_stream_b.argtypes = [vp]
        L.rt_tracer_stream_b.restype = vp
        L.rt_tracer_device_pointer.argtypes = [vp, C.c_int]
        L.rt_tracer_device_pointer.restype = vp
        L.rt_tracer_buffer_bytes.argtypes = [vp, C.c_int]
        L.rt_tracer_buffer_bytes.restype = C.c_size_t
        L.rt_tracer_info.argtypes = [vp, u32p]
        L.rt_tracer_last_error.argtypes = [vp]
        L.rt_tracer_last_error.restype = C.c_char_p
        L.rt_last_error.restype = C.c_char_p
        L.rt_device_count.restype = C.c_int
        L.rt_version.restype = C.c_char_p
        L.rt_dbg_hit_triangle.argtypes = [C.c_int, C.c_uint32, C.c_uint32, f32p, f32p, C.c_int,
                                          C.POINTER(C.c_int32), f32p, f32p, f32p]
        L.rt_dbg_sincos.argtypes = [C.c_int, C.c_uint32, f32p, f32p, f32p]
        L.rt_dbg_uniform.argtypes = [C.c_int, C.c_uint32, C.c_uint32, u32p, f32p]
        L.rt_dbg_valu_peak.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rt_dbg_trace_occupancy.argtypes = [C.c_int, C.c_int, C.c_uint32]
        L.rt_dbg_get_ray.argtypes = [vp, C.c_uint32, u32p, u32p, f32p]
        L.rt_dbg_rng_init_host.argtypes = [C.c_uint64, C.c_uint64, u32p]
        L.rt_dbg_rng_init_host.restype = None
        L.rt_tracer_create_multi.argtypes = [u32p, f32p, f32p, C.c_float, C.c_float, C.c_float, C.POINTER(Options),
                                             C.POINTER(C.c_int32), C.c_uint32, C.POINTER(vp)]
        L.rt_group_unique_id.argtypes = [C.c_char_p]
        L.rt_tracer_join_group.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_char_p]
        L.rt_tracer_leave_group.argtypes = [vp]
        L.rt_tracer_gather_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
        L.rt_tracer_band_count.argtypes = [vp]
        L.rt_tracer_band_info.argtypes = [vp, C.c_uint32, u32p]
        L.rt_tracer_join_group_bands.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_char_p, u32p]
        L.rt_balance_rows.argtypes = [C.c_uint32, u32p, C.POINTER(C.c_double), C.c_uint32, u32p]
        L.rt_tracer_rebalance.argtypes = [vp]
        L.rt_tracer_set_band.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.rt_dbg_read_tile_lists.argtypes = [vp, u32p, C.c_size_t, u32p]
        L.rt_dbg_wave_list_counts.argtypes = [vp, C.c_int, u32p, C.c_size_t, u32p, u32p]
        L.rt_dbg_focal_boxes.argtypes = [vp, C.c_float, f32p, C.c_size_t, f32p, C.c_size_t]
        L.rt_tracer_gather_only.argtypes = [vp]
        L.rt_tracer_group_info.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.rt_dbg_classify.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, u32p, C.c_uint32, f32p, C.c_size_t]
        _lib = L
        return _lib


def device_count():
    return int(load_library().rt_device_count())


def balance_rows(row_begin, cost, granule=8):
    """Boundaries (len(cost) + 1 row indices) that equalise the bands' cost (rt_balance_rows)."""
    b = np.ascontiguousarray(row_begin, np.uint32)
    c = np.ascontiguousarray(cost, np.float64)
    out = np.zeros_like(b)
    rc = load_library().rt_balance_rows(c.size, _u32p(b), c.ctypes.data_as(C.POINTER(C.c_double)), granule, _u32p(out))
    if rc != 0:
        raise RtError("rt_balance_rows: invalid partition")
    return [int(x) for x in out]


def group_unique_id():
    """The id (bytes) rank 0 creates and the launcher's rendezvous hands to every rank for JoinGroup."""
    buf = C.create_string_buffer(GROUP_ID_BYTES)
    rc = load_library().rt_group_unique_id(buf)
    if rc != 0:
        raise RtError("rt_group_unique_id failed (%d): %s" % (rc, load_library().rt_last_error().decode()))
    return buf.raw


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class RayTracer:
    """rt::RayTracer.  Reference methods keep their names: Trace, Stop, Resize,
    SetCameraParameters, RotateCamera, UploadScene, SetUpdateCallback, SetFinishedCallback.
    Callbacks receive (image, size_bytes): image is a host numpy view (rows, W) of BGRA8
    uint32 valid during the call (the reference hands a device pointer: PBO interop is cut)."""

    def __init__(self, imageSize, cameraPosition=(0.0, 0.0, 0.0), cameraAngles=(0.0, 0.0), fov=70.0,
                 focalLength=10.0, aperture=4.0, *, seed=None, device=0, math_mode=MATH_FMA,
                 full_height=0, row_begin=0, no_filter=False, no_binning=False, nearest_hit=False, smooth_normals=False, no_macro_bins=False, no_super_bins=False, samples_in_flight=0,
                 lds_chunk=0, bin_list=0, devices=None, no_sure_hit=False, transport="rccl"):
        """devices: a list of HIP device ordinals, one per row band -> the frame is sharded over them inside
        this process (rt_tracer_create_multi; ordinals may repeat); None -> one tracer on `device`."""
        self._lib = load_library()
        self._h = C.c_void_p()
        self._cbs = {}
        self._retired = []
        size = np.array(imageSize, np.uint32)
        opt = Options()
        opt.struct_size = C.sizeof(Options)
        opt.device = device
        opt.full_height, opt.row_begin = full_height, row_begin
        opt.use_time_seed = 1 if seed is None else 0          # Random.cu:45 when no seed is given
        opt.seed = 0 if seed is None else int(seed)
        opt.math_mode = math_mode
        opt.flags = ((FLAG_NO_FILTER if no_filter else 0) | (FLAG_NO_BINNING if no_binning else 0) |
                     (FLAG_NEAREST_HIT if nearest_hit else 0) | (FLAG_SMOOTH_NORMALS if smooth_normals else 0) | (FLAG_NO_MACRO_BINS if no_macro_bins else 0) | (FLAG_NO_SUPER_BINS if no_super_bins else 0) |
                     (FLAG_NO_SURE_HIT if no_sure_hit else 0))
        opt.samples_in_flight, opt.lds_chunk, opt.bin_list = samples_in_flight, lds_chunk, bin_list
        opt.transport = {"rccl": TRANSPORT_RCCL, "peer": TRANSPORT_PEER}[transport]
        if devices is not None:
            devs = np.ascontiguousarray(devices, np.int32)
            rc = self._lib.rt_tracer_create_multi(_u32p(size), _f32p(np.array(cameraPosition, np.float32)),
                                                  _f32p(np.array(cameraAngles, np.float32)), fov, focalLength, aperture,
                                                  C.byref(opt), devs.ctypes.data_as(C.POINTER(C.c_int32)), devs.size,
                                                  C.byref(self._h))
        else:
            rc = self._lib.rt_tracer_create_ex(_u32p(size), _f32p(np.array(cameraPosition, np.float32)),
                                               _f32p(np.array(cameraAngles, np.float32)), fov, focalLength,
                                               aperture, C.byref(opt), C.byref(self._h))
        if rc != 0 or not self._h:
            raise RtError("rt_tracer_create failed (%d): %s" % (rc, self._lib.rt_last_error().decode()))
        self.full_height = int(full_height) if full_height else int(size[1])
        self._band = bool(full_height) and devices is None
        self.width = int(size[0])
        self.rows = int(size[1])

    # ---- reference API -------------------------------------------------------------
    def Trace(self, iterationCount, samplesPerIteration, updateInterval):
        self._check(self._lib.rt_tracer_trace(self._h, iterationCount, samplesPerIteration, updateInterval))

    def Stop(self):
        self._lib.rt_tracer_stop(self._h)

    def Resize(self, size):
        s = np.array(size, np.uint32)
        self._check(self._lib.rt_tracer_resize(self._h, _u32p(s)))
        self.width, self.rows = int(s[0]), int(s[1])
        if not self._band:                      # whole-frame and multi-device handles: the frame IS the new size
            self.full_height = self.rows

    def SetCameraParameters(self, fov, focalLength, aperture):
        self._lib.rt_tracer_set_camera_parameters(self._h, fov, focalLength, aperture)

    def RotateCamera(self, angles):
        self._lib.rt_tracer_rotate_camera(self._h, _f32p(np.array(angles, np.float32)))

    def UploadScene(self, hostData):
        """hostData: (3N, 4) float32.  Like the reference, an invalid size is rejected
        without raising (RayTracerImpl.cu:121-125); returns False in that case."""
        a = np.ascontiguousarray(hostData, np.float32).reshape(-1, 4)
        rc = self._lib.rt_tracer_upload_scene(self._h, a.ctypes.data, a.shape[0])
        if rc == 1:
            return False
        self._check(rc)
        return True

    def UploadSceneEdges(self, hostData):
        """(3N, 4) float32 in the v0, e0, e1 layout of Documentation/gpu.meshes.txt:16-17."""
        a = np.ascontiguousarray(hostData, np.float32).reshape(-1, 4)
        rc = self._lib.rt_tracer_upload_scene_edges(self._h, a.ctypes.data, a.shape[0])
        if rc == 1:
            return False
        self._check(rc)
        return True

    def SetUpdateCallback(self, callback):
        self._set_cb("update", callback, self._lib.rt_tracer_set_update_callback)

    def SetFinishedCallback(self, callback):
        self._set_cb("finished", callback, self._lib.rt_tracer_set_finished_callback)

    # ---- extensions ------------------------------------------------------------------
    def Wait(self):
        ok = bool(self._lib.rt_tracer_wait(self._h))
        self._retired.clear()                  # the render thread has ended: nobody holds a replaced thunk
        return ok

    def SetSeed(self, seed):
        self._check(self._lib.rt_tracer_set_seed(self._h, int(seed)))

    def UploadSpheres(self, spheres):
        a = np.ascontiguousarray(spheres, np.float32).reshape(-1, 4)
        self._check(self._lib.rt_tracer_upload_spheres(self._h, a.ctypes.data, a.shape[0]))

    def TraceEnqueue(self, iterationCount, samplesPerIteration):
        self._check(self._lib.rt_tracer_trace_enqueue(self._h, iterationCount, samplesPerIteration))

    def TraceEnqueueN(self, iterationCount, samplesPerIteration, n_steps):
        """n_steps passes of TraceEnqueue enqueued by one call (the step loop runs inside the library)."""
        self._check(self._lib.rt_tracer_trace_enqueue_n(self._h, iterationCount, samplesPerIteration, n_steps))

    def Launch(self, samples, clear_first=False, emit_image=False, iterations=1):
        """`iterations` iterations of TraceFunct's loop on the device as one launch (no callbacks, no
        host sync); iterations <= FusedIterations(samples)."""
        if iterations == 1:
            self._check(self._lib.rt_tracer_launch(self._h, samples, int(clear_first), int(emit_image)))
        else:
            self._check(self._lib.rt_tracer_launch_iterations(self._h, samples, iterations, int(clear_first), int(emit_image)))

    def SetListReuse(self, across_traces=True):
        """Keep the tiles' candidate lists from one Trace to the next while nothing they depend on changes
        (default) or only within one Trace (False: every Trace classifies afresh)."""
        self._check(self._lib.rt_tracer_set_list_reuse(self._h, 1 if across_traces else 0))

    def SetImageMirror(self, device_visible_ptr):
        """Emitting Launch/TraceEnqueue launches also write the BGRA8 image to this device-visible
        buffer (e.g. a collective's send tensor); 0/None switches it off."""
        self._check(self._lib.rt_tracer_set_image_mirror(self._h, C.c_void_p(device_visible_ptr or 0)))

    def FusedIterations(self, samples):
        """How many consecutive iterations one launch may run (1 = no fusing)."""
        return max(1, int(self._lib.rt_tracer_fused_iterations(self._h, samples)))

    def Sync(self):
        self._check(self._lib.rt_tracer_sync(self._h))

    def TraceStats(self, samples):
        """One instrumented launch; see rt_tracer_trace_stats in include/rt_mi355x.h."""
        out = (C.c_uint64 * 16)()
        self._check(self._lib.rt_tracer_trace_stats(self._h, samples, out))
        v = [int(x) for x in out]
        return {"exit_det": v[0], "exit_u": v[1], "exit_v": v[2], "exit_hit": v[3],
                "skip_a": v[4], "skip_b": v[5], "skip_c": v[6], "reach_d": v[7],
                "bin_candidates": v[8], "bin_rounds": v[9], "pretest_skips": v[10],
                "tiles_by_list": {"0": v[11], "1": v[12], "sure": v[13], "2": v[14], "more": v[15]}}

    def KernelTime(self, reset=True):
        ms, n = C.c_double(), C.c_uint64()
        self._lib.rt_tracer_kernel_time(self._h, C.byref(ms), C.byref(n), 1 if reset else 0)
        return ms.value, n.value

    def LaunchTime(self, reset=True):
        """(total ms, launches) of the sampled launches by their cost: split launches to the end of the later half (rt_tracer_launch_time)."""
        ms, n = C.c_double(), C.c_uint64()
        self._lib.rt_tracer_launch_time(self._h, C.byref(ms), C.byref(n), 1 if reset else 0)
        return ms.value, n.value

    def _read(self, which, dtype, shape):
        out = np.empty(shape, dtype)
        self._check(self._lib.rt_tracer_read_buffer(self._h, which, out.ctypes.data, out.nbytes))
        return out

    def RenderBuffer(self):
        return self._read(BUF_RENDER, np.float32, (self.rows, self.width, 4))

    def SampleCounts(self):
        return self._read(BUF_COUNTS, np.uint32, (self.rows, self.width))

    def Image(self):
        return self._read(BUF_IMAGE, np.uint32, (self.rows, self.width))

    def RngStates(self):
        """(rows, W, 6) uint32 {d, v0..v4} per pixel (device layout is 6 planes)."""
        planes = self._read(BUF_RNG, np.uint32, (6, self.rows, self.width))
        return np.ascontiguousarray(np.moveaxis(planes, 0, -1))

    # ---- a frame sharded over several GPUs ------------------------------------------------------
    def JoinGroup(self, n_ranks, rank, unique_id=None, row_begin=None):
        """This band tracer becomes rank `rank` of `n_ranks` processes that share one frame (collective call).
        row_begin: n_ranks + 1 boundaries of an explicit partition (default: equal bands)."""
        if row_begin is None:
            self._check(self._lib.rt_tracer_join_group(self._h, n_ranks, rank, unique_id))
        else:
            b = np.ascontiguousarray(row_begin, np.uint32)
            self._check(self._lib.rt_tracer_join_group_bands(self._h, n_ranks, rank, unique_id, _u32p(b)))

    def SetBand(self, row_begin, rows):
        """Move a band tracer to other rows of its frame (buffers and RNG states re-created, like Resize)."""
        self._check(self._lib.rt_tracer_set_band(self._h, row_begin, rows))
        self.rows = int(rows)

    def Rebalance(self):
        """Multi-device tracer: re-partition the rows so that every band costs the same, from the bands' kernel times."""
        self._check(self._lib.rt_tracer_rebalance(self._h))

    def LeaveGroup(self):
        self._check(self._lib.rt_tracer_leave_group(self._h))

    def Frame(self):
        """The gathered (H, W) BGRA8 frame: rank 0 of a group / a multi-device tracer."""
        return self._read(BUF_FRAME, np.uint32, (self.full_height, self.width))

    def GatherTime(self, reset=True):
        ms, n = C.c_double(), C.c_uint64()
        self._lib.rt_tracer_gather_time(self._h, C.byref(ms), C.byref(n), 1 if reset else 0)
        return ms.value, n.value

    def GatherOnly(self):
        """One more exchange of the tiles as they are, no tracing (collective in a multi-process group)."""
        self._check(self._lib.rt_tracer_gather_only(self._h))

    def GroupInfo(self):
        """dict: transport, ranks, band -> rank map, local devices, RCCL version and communicators (rt_tracer_group_info)."""
        import json
        buf = C.create_string_buffer(8192)
        self._check(self._lib.rt_tracer_group_info(self._h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def Bands(self):
        """[{device, row0, rows, rank}] of the handle's bands (one entry for a plain tracer)."""
        out = []
        for k in range(int(self._lib.rt_tracer_band_count(self._h))):
            v = np.zeros(4, np.uint32)
            self._check(self._lib.rt_tracer_band_info(self._h, k, _u32p(v)))
            out.append({"device": int(v[0]), "row0": int(v[1]), "rows": int(v[2]), "rank": int(v[3])})
        return out

    def DevicePointer(self, which):
        return self._lib.rt_tracer_device_pointer(self._h, which)

    def CopyToDevice(self, which, dst_ptr, nbytes):
        self._check(self._lib.rt_tracer_copy_buffer_to_device(self._h, which, dst_ptr, nbytes))

    def CopyToDeviceAsync(self, which, dst_ptr, nbytes):
        """Enqueue the copy on the tracer's stream, no host sync (order other streams with Stream())."""
        self._check(self._lib.rt_tracer_copy_buffer_to_device_async(self._h, which, dst_ptr, nbytes))

    def Stream(self):
        """The tracer's hipStream_t as an integer (torch.cuda.ExternalStream(ptr) wraps it)."""
        return int(self._lib.rt_tracer_stream(self._h) or 0)

    def StreamB(self):
        """The second stream, on which the lower half of split trace launches runs (rt_tracer_stream_b)."""
        return int(self._lib.rt_tracer_stream_b(self._h) or 0)

    def Info(self):
        out = np.zeros(8, np.uint32)
        self._lib.rt_tracer_info(self._h, _u32p(out))
        keys = ("samples_in_flight", "lds_chunk", "lds_bytes", "grid_x", "grid_y", "n_tris", "n_spheres", "device")
        return dict(zip(keys, (int(v) for v in out)))

    def LastError(self):
        return self._lib.rt_tracer_last_error(self._h).decode()

    def DebugTileLists(self):
        """(tiles_y, tiles_x) arrays (count, winner triangle, certain-winner flag) of the stored tile lists."""
        bx, by = (self.width + 31) // 32, (self.rows + 7) // 8
        wpt = np.zeros(1, np.uint32)
        buf = np.zeros(bx * (by + 1) * 4 * 1025, np.uint32)
        self._check(self._lib.rt_dbg_read_tile_lists(self._h, _u32p(buf), buf.size, _u32p(wpt)))
        w = int(wpt[0])
        words = buf[:bx * by * 4 * w].reshape(by, bx * 4, w)[:, :, 0]
        return (words & 0x3FF).astype(np.int32), ((words >> 10) & 0x3FF).astype(np.int32), (words >> 31).astype(bool)

    def DebugWaveListCounts(self, half=0):
        """(counts, capacity): candidate count per tile (grid order of the half's launch; 0xFFFFFFFF = overflow) of a dense scene's
        lists in HBM (rt_dbg_wave_list_counts)."""
        n, cap = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
        buf = np.zeros(((self.width + 31) // 32) * ((self.rows + 7) // 8 + 1) * 4, np.uint32)
        self._check(self._lib.rt_dbg_wave_list_counts(self._h, half, _u32p(buf), buf.size, _u32p(n), _u32p(cap)))
        return buf[:int(n[0])].copy(), int(cap[0])

    def DebugTileListWords(self):
        """(tiles_y, tiles_x, 1 + bin_list) words of the stored tile lists: word 0 = count | winner << 10 | certain << 31, then
        the kept triangle indices in ascending order (rt_dbg_read_tile_lists)."""
        bx, by = (self.width + 31) // 32, (self.rows + 7) // 8
        wpt = np.zeros(1, np.uint32)
        buf = np.zeros(bx * (by + 1) * 4 * 1025, np.uint32)
        self._check(self._lib.rt_dbg_read_tile_lists(self._h, _u32p(buf), buf.size, _u32p(wpt)))
        w = int(wpt[0])
        return buf[:bx * by * 4 * w].reshape(by, bx * 4, w).copy()

    def DebugFocalBoxes(self, curv_scale=1.0):
        """(boxes (tiles_y, tiles_x, 8): lo[3], hi[3], corner path, usable; focal (rows, width, 3)) of a trace launch's tiles."""
        bx, by = (self.width + 31) // 32, (self.rows + 7) // 8
        boxes = np.zeros((by, bx * 4, 8), np.float32)
        focal = np.zeros((self.rows, self.width, 3), np.float32)
        self._check(self._lib.rt_dbg_focal_boxes(self._h, float(curv_scale), _f32p(boxes), boxes.size, _f32p(focal), focal.size))
        return boxes, focal

    def DebugClassify(self, regions, level=0, forms=False, slack_milli=1000):
        """rt_dbg_classify: (header (n, 16), records (n, n_tris, 12 | 32)) for the regions [(x0, y0)] of the band."""
        reg = np.ascontiguousarray(regions, np.uint32).reshape(-1, 2)
        n_tris = self.Info()["n_tris"]
        per = 16 + n_tris * (32 if forms else 12)
        out = np.zeros((reg.shape[0], per), np.float32)
        self._check(self._lib.rt_dbg_classify(self._h, level, 1 if forms else 0, slack_milli, _u32p(reg), reg.shape[0], _f32p(out), out.size))
        return out[:, :16], out[:, 16:].reshape(reg.shape[0], n_tris, 32 if forms else 12)

    def DebugGetRay(self, pixels, states):
        pix = np.ascontiguousarray(pixels, np.uint32).reshape(-1, 2)
        st = np.ascontiguousarray(states, np.uint32).reshape(-1, 6).copy()
        rays = np.zeros((pix.shape[0], 6), np.float32)
        self._check(self._lib.rt_dbg_get_ray(self._h, pix.shape[0], _u32p(pix), _u32p(st), _f32p(rays)))
        return rays, st

    def close(self):
        if self._h:
            self._lib.rt_tracer_destroy(self._h)
            self._h = C.c_void_p()
            self._retired.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- internals -------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise RtError("librt_mi355x error %d: %s" % (rc, self.LastError()))

    def _set_cb(self, key, callback, setter):
        # The render thread snapshots the callback per launch and may still hold the previous thunk (the
        # pipelined update hand-off keeps it across the next launch): a replaced thunk is retired, not freed,
        # until no render thread can be running (Wait(), close()).
        old = self._cbs.get(key)
        if old is not None:
            self._retired.append(old)
        if callback is None:
            self._cbs[key] = CALLBACK()
            setter(self._h, self._cbs[key], None)
            return

        me = weakref.ref(self)                 # no cycle self -> _cbs -> thunk -> closure -> self: a tracer that is dropped
                                               # without close() is still freed by its reference count

        def tramp(ptr, size, _user):
            n = size // 4
            img = np.ctypeslib.as_array(ptr, shape=(n,))
            owner = me()
            rows, width = (owner.rows, owner.width) if owner is not None else (0, 0)   # read per call: Resize changes them
            callback(img.reshape(rows, width) if n == rows * width and n else img, size)
        self._cbs[key] = CALLBACK(tramp)       # keep alive
        setter(self._h, self._cbs[key], None)


# ---- single-function device harnesses (parity tests) ------------------------------------
def dbg_hit_triangle(rays, tris, math_mode=MATH_FMA, eps_mode=0, device=0):
    L = load_library()
    r = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    t = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    n = r.shape[0]
    hit = np.zeros(n, np.int32)
    tuv, nrm, pt = (np.zeros((n, 3), np.float32) for _ in range(3))
    rc = L.rt_dbg_hit_triangle(device, math_mode, n, _f32p(r), _f32p(t), eps_mode,
                               hit.ctypes.data_as(C.POINTER(C.c_int32)), _f32p(tuv), _f32p(nrm), _f32p(pt))
    if rc != 0:
        raise RtError("rt_dbg_hit_triangle failed (%d): %s" % (rc, L.rt_last_error().decode()))
    return hit.astype(bool), tuv, nrm, pt


def dbg_sincos(x, device=0):
    L = load_library()
    a = np.ascontiguousarray(x, np.float32).ravel()
    s, c = np.zeros_like(a), np.zeros_like(a)
    rc = L.rt_dbg_sincos(device, a.size, _f32p(a), _f32p(s), _f32p(c))
    if rc != 0:
        raise RtError("rt_dbg_sincos failed (%d): %s" % (rc, L.rt_last_error().decode()))
    return s, c


def dbg_valu_peak(device=0):
    """(attainable lane-FMA per second, shader clock GHz) measured on `device`."""
    L = load_library()
    r, g = C.c_double(), C.c_double()
    rc = L.rt_dbg_valu_peak(device, C.byref(r), C.byref(g))
    if rc != 0:
        raise RtError("rt_dbg_valu_peak failed (%d): %s" % (rc, L.rt_last_error().decode()))
    return r.value, g.value


def dbg_check_midrange(device=0):
    """Exhaustive device check of normalize's mid-range sqrt/reciprocal fast paths:
    (values checked, sqrt mismatches, reciprocal mismatches, a mismatching operand's bits or 0)."""
    L = load_library()
    out = (C.c_uint64 * 4)()
    L.rt_dbg_check_midrange.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
    rc = L.rt_dbg_check_midrange(device, out)
    if rc != 0:
        raise RtError("rt_dbg_check_midrange failed (%d): %s" % (rc, L.rt_last_error().decode()))
    return tuple(int(x) for x in out)


def dbg_uniform(states, m, device=0):
    L = load_library()
    st = np.ascontiguousarray(states, np.uint32).reshape(-1, 6).copy()
    out = np.zeros((st.shape[0], m), np.float32)
    rc = L.rt_dbg_uniform(device, st.shape[0], m, _u32p(st), _f32p(out))
    if rc != 0:
        raise RtError("rt_dbg_uniform failed (%d): %s" % (rc, L.rt_last_error().decode()))
    return out, st


def dbg_rng_init_host(seed, subsequence):
    s = np.zeros(6, np.uint32)
    load_library().rt_dbg_rng_init_host(int(seed), int(subsequence), _u32p(s))
    return s
