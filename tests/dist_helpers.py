"""Test-side stand-ins for the two seams of raytracertest_amd.dist.RowBandJob:

  HostStagedExchange   tiles through host memory and a gloo gather -- for two ranks that share the ONE GPU of a
                       test box (RCCL refuses two ranks on one device) and for the CPU-only tests;
  OracleBackedTracer   the oracle behind the few RayTracer methods RowBandJob calls -- lets the CPU suite run the
                       job's host logic (band geometry, control-plane collectives, progressive loop) without a GPU.

Test infrastructure only: nothing under raytracertest_amd/ imports this."""
import ctypes as C

import numpy as np


class HostStagedExchange:
    def attach(self, job, row_begin=None):
        self._frame = None

    def detach(self, job):
        pass

    def after_emit(self, job):
        import torch
        from raytracertest_amd.dist import gather_tiles
        tile = torch.from_numpy(np.ascontiguousarray(job.tracer.Image()).view(np.int32))
        f = gather_tiles(tile, job.world, job.rank)
        self._frame = None if f is None else f.numpy().view(np.uint32)

    def frame(self, job):
        return self._frame


class OracleBackedTracer:
    def __init__(self, W, full_h, row0, rows, cfg, nthreads=2):
        from oracle import oracle_py as orc
        self._orc = orc
        self.o = orc.OracleTracer(W, full_h, cfg["angles"], cfg["fov"], cfg["focal"], cfg["aperture"], seed=cfg["seed"],
                                  row0=row0, rows=rows, nthreads=nthreads)
        self.launch_log = []

    def UploadScene(self, tris):
        return self.o.upload_scene(tris)

    def UploadSpheres(self, s):
        self.o.upload_spheres(s)

    def TraceEnqueue(self, iterations, samples):
        self.o.trace(iterations, samples)

    def Launch(self, samples, clear_first=False, emit_image=False, iterations=1):
        self.launch_log.append(iterations)
        L = self._orc.lib()
        if clear_first:
            L.orc_frame_clear(C.byref(self.o._frame))
        for _ in range(iterations):
            self.o.launch(samples)
        if emit_image:
            L.orc_convert(C.byref(self.o._frame))

    def FusedIterations(self, samples):
        return max(1, 4 // max(samples, 1))

    def Sync(self):
        pass

    def Image(self):
        return self.o.image

    def SetListReuse(self, on):
        pass

    def KernelTime(self, reset=True):
        return 0.0, 0

    def LaunchTime(self, reset=True):
        return 0.0, 0

    def GatherTime(self, reset=True):
        return 0.0, 0

    def Bands(self):
        return [{"device": -1, "row0": self.o.row0, "rows": self.o.rows, "rank": 0}]

    def close(self):
        pass
