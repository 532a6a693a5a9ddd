"""A stage-1 training iteration as ONE hipGraph launch.

A stage-1 iteration at the reference's shape (train_3d.py:189-250: MPMesh.forward on one 180 x 320 crop, image + loop-mask loss, four
regularisers, backward, Adam) is ~80 kernel launches for 0.9 ms of GPU time: the host, not the GPU, sets its rate (1.1-1.3 ms).  Nothing in
it depends on values the host must read back, so the whole iteration -- forward, loss, backward, optimiser step -- is recorded once
(torch.cuda.CUDAGraph = hipGraph on ROCm) and replayed with ONE launch per iteration.  What changes from iteration to iteration reaches
the recorded kernels through device memory that is refreshed before the replay:

  * the view: the plane homographies [D,3,3], formed on the host from the pose as always (MPMesh.plane_homographies), copied into a static
    device tensor the module's render reads (MPMesh._static_homos);
  * the crop of the training image and of the loop mask: copied into static tensors;
  * the optimiser's scalars (learning rate / bias corrections of THIS step): tiles.TileAdam in device-scalar mode (vl3d_adam_step_tiles_dev).

Measured (round 4, examples/stage1_step.py --graph, D = 32, 576 x 1024 planes, 180 x 320 crops): 970-1030 it/s against 940-1100 eager on the
same box; a whole 720p frame 430 against 470.  After the module's host path was trimmed (fused smoothness terms, no copies for the
single-view batch, numpy homographies) the eager iteration runs at 0.9 ms of GPU time in ~1.0 ms: there is no launch bound left for the
graph to remove, and a hipGraph replay costs about a launch per node.  The recorded iteration is therefore an OPTION (slow hosts, many
small kernels per view), not what the examples or the bench use.

The first `warmup` calls run eagerly (they are real training steps; they also fill the module's small caches and the allocator), the next
call records, every later call replays.  The loss closure must consist of device operators only.  Same arithmetic, same kernels, same
parameters as the eager loop (tests/test_gpu_mpv.py).
"""
import numpy as np
import torch


class GraphedStage1Iteration:
    def __init__(self, model, optimizer, h, w, loss_fn, loop_mask=True, warmup=3):
        """model: MPMesh (dense, CUDA, training mode); optimizer: the tiles.TileAdam of model.get_optimizer(); (h, w): the crop size;
        loss_fn(rgbl [1,C,h,w], extra dict, target [1,3,h,w], target_mask [1,h,w] or None) -> scalar loss."""
        from .tiles import TileAdam
        if not isinstance(optimizer, TileAdam):
            raise RuntimeError("GraphedStage1Iteration drives tiles.TileAdam (MPMesh.get_optimizer() of a CUDA model)")
        self.model, self.opt, self.h, self.w, self.loss_fn, self.warmup = model, optimizer, int(h), int(w), loss_fn, int(warmup)
        dev = model.stack.device
        self.dev = dev
        self.homos = torch.zeros((model.mpi_d, 3, 3), dtype=torch.float32, device=dev)
        # pinned staging for the view: a RING, each slot guarded by an event -- the host runs several iterations ahead of the GPU once the
        # iteration is one launch, and must not overwrite a slot whose asynchronous copy has not run yet
        self._ring = [(torch.zeros((model.mpi_d, 3, 3), dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(16)]
        self.target = torch.zeros((1, 3, self.h, self.w), dtype=torch.float32, device=dev)
        self.target_mask = torch.zeros((1, self.h, self.w), dtype=torch.float32, device=dev) if loop_mask else None
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.graph = None
        self.calls = 0
        self.stream = torch.cuda.Stream(device=dev)
        optimizer.use_device_scalars(dev)
        self._eye = (torch.eye(4, dtype=torch.float64)[None], torch.eye(3, dtype=torch.float64)[None])

    def _iteration(self):
        rgbl, extra = self.model(self.h, self.w, *self._eye)          # (the view comes from self.homos)
        loss = self.loss_fn(rgbl, extra, self.target, self.target_mask)
        loss.backward()
        self.opt.step()
        self.loss.copy_(loss.detach())

    def __call__(self, tar_extrin, tar_intrin, target, target_mask=None):
        """one training iteration for the view (tar_extrin [1,4,4], tar_intrin [1,3,3]: HOST tensors, as the DataLoader yields them) and the
        crop `target` [1,3,h,w] (+ `target_mask` [1,h,w]) on the device.  -> the loss (a device scalar that the next call overwrites)."""
        m = self.model
        tar_extrin, tar_intrin = torch.as_tensor(tar_extrin), torch.as_tensor(tar_intrin)
        extrin = tar_extrin @ m._on(tar_extrin.device, "ref_extrin")[None, ...].inverse().to(tar_extrin.dtype)      # MPI.py:596 (as MPMesh.forward)
        pin, ev = self._ring[self.calls % len(self._ring)]
        if self.calls >= len(self._ring):
            ev.synchronize()
        pin.copy_(m.plane_homographies(extrin[:1], tar_intrin[:1]))
        cur = torch.cuda.current_stream(self.dev)
        self.stream.wait_stream(cur)                 # (the caller produced `target` on its stream)
        with torch.cuda.stream(self.stream):
            self.homos.copy_(pin, non_blocking=True)
            ev.record(self.stream)
            self.target.copy_(target, non_blocking=True)
            if self.target_mask is not None:
                self.target_mask.copy_(target_mask, non_blocking=True)
            self.opt.prepare_step()
            m._static_homos = self.homos
            try:
                if self.calls < self.warmup:
                    self.opt.zero_grad(set_to_none=True)
                    self._iteration()
                else:
                    if self.graph is None:
                        self.opt.zero_grad(set_to_none=True)
                        self.graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(self.graph, stream=self.stream):
                            self._iteration()
                    self.graph.replay()
            finally:
                m._static_homos = None
        cur.wait_stream(self.stream)                 # (the caller reads the loss / the parameters on its stream)
        self.calls += 1
        return self.loss

    def synchronize(self):
        self.stream.synchronize()
