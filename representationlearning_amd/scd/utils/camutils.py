"""Multi-scale class-activation maps (reference: SCD-AAAI2023/utils/camutils.py:85-148).  Per scale: one batched forward of the images
and their horizontal flips, then ONE launch that samples both low-resolution maps at every image pixel (align_corners=False),
un-flips, takes the maximum, applies the ReLU and adds into the running sum; one launch normalises the planes at the end.  The
reference builds five full-resolution intermediates per scale for the same result."""
import torch

from ... import ops


def _scales(scales):
    return [1.0] + [s for s in scales if s != 1.0]          # the reference always evaluates scale 1 first (:89-97), then the others


def _cam_pass(model, inputs, s):
    b, c, h, w = inputs.shape
    x = inputs if s == 1.0 else ops.resize_bilinear_planar(inputs.float(), (int(s * h), int(s * w)))
    return model(torch.cat([x, x.flip(-1)], dim=0), cam_only=True)


def _merge(acc, cam, first):
    camh = cam.permute(0, 2, 3, 1)
    return ops.cam_merge_(acc, camh if camh.is_contiguous() else camh.contiguous(), accumulate=not first)


def multi_scale_cam(model, inputs, scales):
    b, c, h, w = inputs.shape
    acc = None
    with torch.no_grad():
        for i, s in enumerate(_scales(scales)):
            cam, _ = _cam_pass(model, inputs, s)
            if acc is None:
                acc = torch.empty(b, cam.shape[1], h, w, device=inputs.device, dtype=torch.float32)
            _merge(acc, cam, i == 0)
        return ops.cam_normalize_(acc)


def multi_scale_cam_with_ref_mat(model, inputs, scales):
    """As multi_scale_cam, also returning the attention prediction `ref_mat[argmax(scales)]` (:115-148; the list is in evaluation
    order - scale 1 first - and is indexed by the position of the largest entry of `scales`, as the reference does)."""
    b, c, h, w = inputs.shape
    acc, ref_mat = None, []
    with torch.no_grad():
        for i, s in enumerate(_scales(scales)):
            cam, ref = _cam_pass(model, inputs, s)
            ref_mat.append(ref)
            if acc is None:
                acc = torch.empty(b, cam.shape[1], h, w, device=inputs.device, dtype=torch.float32)
            _merge(acc, cam, i == 0)
        return ops.cam_normalize_(acc), ref_mat[max(range(len(scales)), key=lambda j: scales[j])]


class GraphedMultiScaleCam:
    """multi_scale_cam for a FIXED input shape as one replayed hipGraph: at the reference's batch of two the six forwards are a few
    hundred short launches and the call is bound by issuing them.  The packed weights are formed during the warm-up calls (outside
    the capture) and stay cached until a parameter changes; the object holds the packed buffers its graph reads and refuses to replay once a
    parameter's address or version counter differs from the capture's - re-create it after loading new weights.

        cam_fn = GraphedMultiScaleCam(model, inputs, scales);  cams = cam_fn(inputs)        # cams: a buffer the next call overwrites
    """

    def __init__(self, model, example, scales, autocast_dtype=None, warmup=2):
        self.scales, self.dtype = list(scales), autocast_dtype
        self.static_in = example.detach().clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                        # capture needs prior work on a non-default stream
            for _ in range(warmup):
                self._run(model)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = self._run(model)
        # the graph reads the packed weights the warm-up calls cached on the convolutions: hold them (a later re-pack replaces the
        # cache entry and would free the buffer under the graph), and remember which parameter values they were packed from
        from ... import nnf
        self._held = nnf.packed_weights(model)
        self._params = list(model.parameters())
        self._tags = self._param_tags()

    def _param_tags(self):
        return [(p.data_ptr(), p._version) for p in self._params]

    def _run(self, model):
        with torch.autocast("cuda", dtype=self.dtype or torch.bfloat16, enabled=self.dtype is not None):
            return multi_scale_cam(model, self.static_in, self.scales)

    def __call__(self, inputs):
        if inputs.shape != self.static_in.shape:
            raise ValueError(f"GraphedMultiScaleCam was captured for {tuple(self.static_in.shape)}, got {tuple(inputs.shape)}")
        if self._param_tags() != self._tags:
            raise RuntimeError("GraphedMultiScaleCam: the model's parameters changed after the capture (load_state_dict / an optimizer "
                               "step) - the graph holds the weights packed at capture time; create a new GraphedMultiScaleCam")
        self.static_in.copy_(inputs)
        self.graph.replay()
        return self.static_out
