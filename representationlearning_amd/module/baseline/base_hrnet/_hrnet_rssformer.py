"""HRNetV2 backbone with one GeneralTransformerBlock per HighResolutionModule
(reference: module/baseline/base_hrnet/_hrnet_rssformer.py:216-705).

Same module tree / attribute names as the reference, hence identical state_dict keys.  Activations are kept in
channels-last memory (NHWC physically) end to end — the layout the HIP kernels and the window attention want.
Every Conv2d -> BatchNorm2d (-> ReLU / + residual) group runs as one fused autograd node on the hand-written
implicit-GEMM / BN kernels (representationlearning_amd.nnf), nearest upsampling is fused with the branch sum;
the remaining branch adds / ReLU are ATen elementwise kernels.  nn.Conv2d / nn.BatchNorm2d modules only hold the parameters."""
import contextlib
import os

import torch
import torch.nn as nn

from .... import nnf
from .modules.MTFM import GeneralTransformerBlock

BatchNorm2d = nn.BatchNorm2d
BN_MOMENTUM = 0.1

__all__ = ["HighResolutionNet", "HighResolutionModule", "hrnetv2_w18", "hrnetv2_w32", "hrnetv2_w40", "hrnetv2_w48"]


def _table(c):
    """Stage configuration for branch widths c = (c0, c1, c2, c3)."""
    return dict(
        stage1=dict(num_modules=1, num_branches=1, block="BOTTLENECK", num_blocks=(4,), num_channels=(64,), fuse_method="SUM"),
        stage2=dict(num_modules=1, num_branches=2, block="BASIC", num_blocks=(4, 4), num_channels=c[:2], fuse_method="SUM"),
        stage3=dict(num_modules=4, num_branches=3, block="BASIC", num_blocks=(4, 4, 4), num_channels=c[:3], fuse_method="SUM"),
        stage4=dict(num_modules=3, num_branches=4, block="BASIC", num_blocks=(4, 4, 4, 4), num_channels=c[:4], fuse_method="SUM"))


# widths of the reference's model_extra (:38-184); 'hrnetv2_w18' is a KeyError there (renamed 'hrnetv2_w32s', :39),
# here both names resolve to the 18-wide table ("Tiny", SURVEY §8 variants).
model_extra = {
    "hrnetv2_w32s": _table((18, 36, 72, 144)),
    "hrnetv2_w18": _table((18, 36, 72, 144)),
    "hrnetv2_w32": _table((32, 64, 128, 256)),
    "hrnetv2_w40": _table((40, 80, 160, 320)),
    "hrnetv2_w48": _table((48, 96, 192, 384)),
}


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


_FANOUT_SIDE = os.environ.get("RSSF_FANOUT_SIDE", "1") != "0"      # A/B switch


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, producer=None, want_link=False):
        """producer: nnf.BnBwdLink of the layer whose output x is, when this block is that output's only consumer (run_blocks);
        want_link: also return the link of bn2 for the next block."""
        res = x if self.downsample is None else nnf.run_sequential(self.downsample, x)
        link = nnf.residual_link(x, res)          # the skip gradient rides on conv1's data-gradient launch
        # BatchNorm-backward statistics come out of the consumer's data-gradient launch: bn1's out of conv2's; the producer's out
        # of conv1's, whose launch (with the skip gradient as addend) forms the COMPLETE gradient of x when `link` is active
        l1 = nnf.bwd_stats_link()
        # relu(bn1(conv1(x))) is consumed by conv2 alone: where conv2's kernels can apply bn1 + ReLU while they stage conv1's raw
        # output, that activation is never written (nnf.can_defer_apply)
        defer = l1 is not None and nnf.can_defer_apply(x, self.conv1, self.conv2)
        out = nnf.conv_bn_act(x, self.conv1, self.bn1, nnf.ACT_RELU, grad_sink=link, stats_out=l1,
                              stats_in=producer if link is not None else None, defer_apply=defer)
        l2 = nnf.bwd_stats_link() if want_link else None
        y = nnf.conv_bn_act(out, self.conv2, self.bn2, nnf.ACT_RELU, res_pre=res, grad_deposit=link, stats_out=l2, stats_in=l1)
        return (y, l2) if want_link else y


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = BatchNorm2d(planes * self.expansion, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, producer=None, want_link=False):
        if self.downsample is None:
            res = x
        else:
            # x feeds the projection AND conv1: one alias each, their gradients meet in the fan-out node's launch (not in autograd's)
            (xd, x), _ = nnf.fanout(x, 0, n_alias=2)
            res = nnf.run_sequential(self.downsample, xd)
        link = nnf.residual_link(x, res)
        l1, l2 = nnf.bwd_stats_link(), nnf.bwd_stats_link()
        out = nnf.conv_bn_act(x, self.conv1, self.bn1, nnf.ACT_RELU, grad_sink=link, stats_out=l1,
                              stats_in=producer if link is not None else None)
        out = nnf.conv_bn_act(out, self.conv2, self.bn2, nnf.ACT_RELU, stats_out=l2, stats_in=l1)
        l3 = nnf.bwd_stats_link() if want_link else None
        y = nnf.conv_bn_act(out, self.conv3, self.bn3, nnf.ACT_RELU, res_pre=res, grad_deposit=link, stats_out=l3, stats_in=l2)
        return (y, l3) if want_link else y


blocks_dict = {"BASIC": BasicBlock, "BOTTLENECK": Bottleneck}


def run_blocks(seq, x):
    """nn.Sequential of residual blocks, block by block: the output of a block is consumed by the next block only, so the
    BatchNorm-backward statistics of its last BatchNorm ride on the next block's first data-gradient launch (nnf.BnBwdLink)."""
    mods = list(seq)
    link = None
    for k, m in enumerate(mods):
        if isinstance(m, (BasicBlock, Bottleneck)):
            if k + 1 < len(mods):
                x, link = m(x, producer=link, want_link=True)
            else:
                x, link = m(x, producer=link), None
        else:
            x, link = m(x), None
    return x


class BlockChain(nn.Sequential):
    """nn.Sequential of residual blocks (same parameters / state_dict keys) whose forward is run_blocks."""

    def forward(self, x):
        return run_blocks(self, x)


def _down_chain(cin, cout, steps):
    """`steps` stride-2 3x3 conv+BN stages; ReLU after all but the last; width changes at the last stage."""
    seq = []
    for k in range(steps):
        last = k == steps - 1
        co = cout if last else cin
        layers = [nn.Conv2d(cin, co, 3, 2, 1, bias=False), BatchNorm2d(co, momentum=BN_MOMENTUM)]
        if not last:
            layers.append(nn.ReLU(False))
        seq.append(nn.Sequential(*layers))
    return nn.Sequential(*seq)


class HighResolutionModule(nn.Module):
    def __init__(self, num_branches, blocks, num_blocks, num_inchannels, num_channels, fuse_method,
                 multi_scale_output=True):
        super().__init__()
        self._check_branches(num_branches, blocks, num_blocks, num_inchannels, num_channels)
        self.num_inchannels = num_inchannels
        self.fuse_method = fuse_method
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        self.branches = self._make_branches(num_branches, blocks, num_blocks, num_channels)
        self.fuse_layers = self._make_fuse_layers()
        self.relu = nn.ReLU(False)
        self.transformer = GeneralTransformerBlock(num_channels[0], planes=num_channels[0], num_heads=2)

    def _check_branches(self, num_branches, blocks, num_blocks, num_inchannels, num_channels):
        for name, seq in (("NUM_BLOCKS", num_blocks), ("NUM_CHANNELS", num_channels), ("NUM_INCHANNELS", num_inchannels)):
            if num_branches != len(seq):
                raise ValueError("NUM_BRANCHES({}) <> {}({})".format(num_branches, name, len(seq)))

    def _make_one_branch(self, i, block, num_blocks, num_channels, stride=1):
        width = num_channels[i] * block.expansion
        downsample = None
        if stride != 1 or self.num_inchannels[i] != width:
            downsample = nn.Sequential(nn.Conv2d(self.num_inchannels[i], width, kernel_size=1, stride=stride, bias=False),
                                       BatchNorm2d(width, momentum=BN_MOMENTUM))
        layers = [block(self.num_inchannels[i], num_channels[i], stride, downsample)]
        self.num_inchannels[i] = width
        layers += [block(width, num_channels[i]) for _ in range(1, num_blocks[i])]
        return BlockChain(*layers)

    def _make_branches(self, num_branches, block, num_blocks, num_channels):
        return nn.ModuleList([self._make_one_branch(i, block, num_blocks, num_channels) for i in range(num_branches)])

    def _make_fuse_layers(self):
        if self.num_branches == 1:
            return None
        ch = self.num_inchannels
        rows = []
        for i in range(self.num_branches if self.multi_scale_output else 1):
            row = []
            for j in range(self.num_branches):
                if j > i:
                    row.append(nn.Sequential(nn.Conv2d(ch[j], ch[i], 1, 1, 0, bias=False),
                                             BatchNorm2d(ch[i], momentum=BN_MOMENTUM),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode="nearest")))
                elif j == i:
                    row.append(None)
                else:
                    row.append(_down_chain(ch[j], ch[i], i - j))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def _transformer_relu(self, low, high):
        """relu(transformer(low, high)) (:430-435): the ReLU rides in the block's last BatchNorm pass (no clamp / mask launches)."""
        if isinstance(self.transformer, GeneralTransformerBlock) and isinstance(self.relu, nn.ReLU):
            return self.transformer(low, high, post_relu=True)
        return self.relu(self.transformer(low, high))

    def get_num_inchannels(self):
        return self.num_inchannels

    def _lockstep_ok(self):
        """Walk the branches block by block (all branches of a module run the same BasicBlock step on independent tensors): every
        phase of a step - convolutions, BatchNorm passes, weight / data gradients - is then ONE grouped launch over the branches
        (nnf.conv_bn_act_group) instead of one latency-bound launch per branch on its own stream, and under data parallelism the
        BatchNorm statistics of all branches (and of all fuse paths of one depth) travel in ONE collective on one communicator."""
        if self.num_branches < 2 or os.environ.get("RSSF_LOCKSTEP", "1") == "0":
            return False
        nblk = len(self.branches[0])
        return all(len(b) == nblk and all(isinstance(m, BasicBlock) and m.downsample is None for m in b) for b in self.branches)

    def _branches_lockstep(self, xs, which=None):
        """The branches `which` (default: all) block by block as groups; xs: their inputs, in that order."""
        xs = list(xs)
        which = list(range(self.num_branches)) if which is None else list(which)
        nb = len(xs)
        if nb == 1:
            return [self.branches[which[0]](xs[0])]
        nblk = len(self.branches[which[0]])
        prev = None                               # GroupBwdLink of the previous block's bn2 group (see run_blocks / nnf.BnBwdLink)
        for k in range(nblk):
            blks = [self.branches[i][k] for i in which]
            links = [nnf.residual_link(xs[i], xs[i]) for i in range(nb)]
            g1 = nnf.group_stats_link(nb)
            # relu(bn1(conv1(x))) has ONE consumer, conv2: where conv2's kernels can apply bn1 + ReLU on load it is never written
            defer = [g1 is not None and nnf.can_defer_apply(xs[i], blks[i].conv1, blks[i].conv2) for i in range(nb)]
            mid = nnf.conv_bn_act_group([dict(x=xs[i], conv=blks[i].conv1, bn=blks[i].bn1, act=nnf.ACT_RELU, grad_sink=links[i], defer_apply=defer[i])
                                         for i in range(nb)], stats_out=g1, stats_in=prev if all(l is not None for l in links) else None)
            g2 = nnf.group_stats_link(nb) if k + 1 < nblk else None
            xs = nnf.conv_bn_act_group([dict(x=mid[i], conv=blks[i].conv2, bn=blks[i].bn2, act=nnf.ACT_RELU, res_pre=xs[i],
                                             grad_deposit=links[i]) for i in range(nb)], stats_out=g2, stats_in=g1)
            prev = g2
        return xs

    def _fuse_lockstep(self, x):
        """The fuse layers (:361-405, 424-435 of the reference) depth by depth: depth 0 holds every 1x1 (j > i) convolution and the
        first stride-2 convolution of every down-sampling chain (j < i), depth d the d-th convolution of the chains that are that
        long.  The chain from branch 0 ends in `relu(bn(conv(.)) + low_i)`; its last convolution waits for the depth at which
        `low_i` is complete (depth max(i - 1, 1)).  Sums are formed in the reference's order (j ascending)."""
        nb, nout = self.num_branches, len(self.fuse_layers)
        x, accs, x0 = self._fanout(x)                # x0: aliases of x[1..] for the 1x1 convolutions towards output 0
        up, cur, done = {}, {}, {}                # (i,j) -> 1x1 output / running chain value / finished chain value
        # depth at which `low_i` is complete: the 1x1 / first-stage outputs exist after depth 0, the chain from branch j >= 1 after
        # depth i - j - 1 (only output 1 of a 2-branch module needs neither: low_1 = x[1])
        last_depth = {i: max(i - 1, 1 if (i < nb - 1 or i > 1) else 0) for i in range(1, nout)}
        lows, outs = {}, [None] * nout

        def low_of(i):
            terms, scales = [], []
            for j in range(1, nb):
                terms.append(x[j] if j == i else up[(i, j)] if j > i else done[(i, j)])
                scales.append(int(self.fuse_layers[i][j][2].scale_factor) if j > i else 1)
            return nnf.fuse_sum(terms, scales)

        def fuse_rest():
            depth = 0
            while True:
                items, tags = [], []
                for i in range(1, nout):
                    for j in range(nb):
                        if j > i and depth == 0:
                            fl = self.fuse_layers[i][j]
                            items.append(dict(x=x[j], conv=fl[0], bn=fl[1], act=nnf.ACT_NONE, grad_accum=accs[j]))
                            tags.append(("up", i, j))
                        elif j < i:
                            chain = self.fuse_layers[i][j]
                            length = len(chain)
                            final0 = j == 0                           # ends in + low_i, ReLU
                            d = depth
                            if final0 and depth == last_depth[i] and (length - 1) <= depth:
                                d = length - 1                        # the (possibly delayed) last convolution of the branch-0 chain
                            elif final0 and depth >= length - 1:
                                continue
                            elif not final0 and depth >= length:
                                continue
                            stage = chain[d]
                            src = x[j] if d == 0 else cur[(i, j)]
                            acc = accs[j] if d == 0 else None
                            if final0 and d == length - 1:
                                if depth != last_depth[i]:
                                    continue
                                if i not in lows:
                                    lows[i] = low_of(i)
                                items.append(dict(x=src, conv=stage[0], bn=stage[1], act=nnf.ACT_RELU, res_pre=lows[i], grad_accum=acc))
                                tags.append(("out", i, j))
                            else:
                                relu = len(stage) > 2
                                items.append(dict(x=src, conv=stage[0], bn=stage[1], act=nnf.ACT_RELU if relu else nnf.ACT_NONE, grad_accum=acc))
                                tags.append(("last" if d == length - 1 else "mid", i, j))
                if not items:
                    break
                res = nnf.conv_bn_act_group(items)
                for (kind, i, j), t in zip(tags, res):
                    if kind == "up":
                        up[(i, j)] = t
                    elif kind == "mid":
                        cur[(i, j)] = t
                    elif kind == "last":
                        done[(i, j)] = t
                    else:
                        outs[i] = t
                depth += 1
            return outs[1:]

        def fuse_zero():
            # output 0: its 1x1 convolutions (one group, one exchange), the upsampled sum, then the transformer block.  No gradient
            # accumulator here: these consumers of x[j] run on the main stream, the accumulating ones of outputs 1.. on the side stream
            its = [dict(x=x0[j], conv=self.fuse_layers[0][j][0], bn=self.fuse_layers[0][j][1], act=nnf.ACT_NONE) for j in range(1, nb)]
            for j, t in zip(range(1, nb), nnf.conv_bn_act_group(its)):
                up[(0, j)] = t
            return self._transformer_relu(low_of(0), x[0])

        # as in the single-GPU path the outputs 1.. run beside output 0's transformer block, on a side stream whose SyncBN exchanges
        # travel on a communicator of their own (nnf.fork_side / Runtime.comm_side); without one, both run on this stream
        rest, y0 = nnf.fork_side(fuse_rest, fuse_zero, x)
        return [y0] + list(rest)

    def forward(self, x):
        if self.num_branches == 1:
            return [self.branches[0](x[0])]
        if self._lockstep_ok():
            nb = self.num_branches
            split = os.environ.get("RSSF_LOCKSTEP_SPLIT", "0")
            if ";" in split:                       # per branch count: "<2 branches>;<3 branches>;<4 branches>"
                split = (split.split(";") + ["all"] * 3)[nb - 2]
            # (one stream only - eager launches, a single SyncBN communicator: all branches in ONE group, the fewest launches / exchanges)
            if split not in ("", "all") and nnf.can_fork_side(x[0]):
                # the branches named in RSSF_LOCKSTEP_SPLIT on the step's stream, the others as lock-step groups beside them
                main = sorted({int(v) for v in split.split(",") if int(v) < nb})
                side = [i for i in range(nb) if i not in main]
                if main and side:
                    ys_side, ys_main = nnf.fork_side(lambda: self._branches_lockstep([x[i] for i in side], side),
                                                     lambda: self._branches_lockstep([x[i] for i in main], main), [x[i] for i in side])
                    ys = [None] * nb
                    for i, y in zip(main, ys_main):
                        ys[i] = y
                    for i, y in zip(side, ys_side):
                        ys[i] = y
                    return self._fuse_lockstep(ys)
            return self._fuse_lockstep(self._branches_lockstep(x[:nb]))
        x = nnf.parallel_map(list(self.branches), x[:self.num_branches])       # BlockChains of BasicBlocks, one stream each
        x, accs, x0 = self._fanout(x)

        def fuse_output(i):
            terms, scales = [], []
            for j in range(1, self.num_branches):
                if j == i:
                    terms.append(x[j]); scales.append(1)
                elif j > i:      # 1x1 conv + BN; its nearest upsample rides in the fuse sum
                    fl = self.fuse_layers[i][j]
                    terms.append(nnf.conv_bn_act(x[j] if i > 0 else x0[j], fl[0], fl[1], grad_accum=accs[j] if i > 0 else None))
                    scales.append(int(fl[2].scale_factor))
                else:
                    terms.append(nnf.run_sequential(self.fuse_layers[i][j], x[j], grad_accum=accs[j])); scales.append(1)
            low = nnf.fuse_sum(terms, scales)
            if i == 0:
                return self._transformer_relu(low, x[0])     # residual comes from `low`; x[0] only feeds K/V (:430-431)
            # relu(fuse[i][0](x[0]) + low) (:432-435): sum and ReLU ride in the last down-sampling conv's BatchNorm pass
            return nnf.run_sequential(self.fuse_layers[i][0], x[0], res_pre=low, act_last=nnf.ACT_RELU, grad_accum=accs[0])

        # Output 0 ends in the transformer block (attention + MlpDWBN at full resolution: the longest serial chain of the module); the
        # other outputs - ten small down-sampling convolutions and their sums - depend on the branch outputs only and run beside it
        # on a side stream (their accumulating consumers share that stream: _fanout counts them alone).
        rest, y0 = nnf.fork_side(lambda: [fuse_output(i) for i in range(1, len(self.fuse_layers))], lambda: fuse_output(0), x)
        return [y0] + list(rest)

    def _fanout(self, x):
        """Branch output j feeds one convolution per fuse path (i != j): their data gradients accumulate in one buffer
        (nnf.GradAccum) instead of being summed by autograd with an elementwise kernel per consumer.  Branch outputs 1.. have two
        more consumers - their own fuse sum (`low_j`, an identity) and the 1x1 convolution towards output 0 (main stream, no
        accumulator): each gets its own alias (second list), and the three gradients meet in the fan-out node's ONE launch."""
        xs, accs, xs0 = [], [], []
        for j in range(self.num_branches):
            n_acc = sum(1 for i in range(1, len(self.fuse_layers)) if i != j)      # (path 0 runs on another stream)
            # The node of a branch output 1.. is created under the SIDE stream (a view: no launch), so its backward - the sum of the
            # accumulated gradients, the fuse sum's and path 0's - runs there: two of its three inputs and its consumer (the
            # branch's backward) live on that stream, only path 0's gradient changes streams (RSSF_FANOUT_SIDE=0: on this stream)
            s = nnf.side_stream(x[j]) if (j >= 1 and _FANOUT_SIDE) else None
            with (torch.cuda.stream(s) if s is not None else contextlib.nullcontext()):
                t, acc = nnf.fanout(x[j], n_acc, n_alias=2 if j >= 1 else 1)
            if j >= 1:
                t, t0 = t
            else:
                t0 = t
            xs.append(t)
            xs0.append(t0)
            accs.append(acc)
        return xs, accs, xs0


class HighResolutionNet(nn.Module):
    def __init__(self, extra, norm_eval=True, zero_init_residual=False, frozen_stages=-1):
        super().__init__()
        self.norm_eval, self.frozen_stages, self.zero_init_residual, self.extra = norm_eval, frozen_stages, zero_init_residual, extra
        self.conv1 = nn.Conv2d(3, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn1 = BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(64, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn2 = BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)

        self.stage1_cfg = extra["stage1"]
        block = blocks_dict[self.stage1_cfg["block"]]
        c1 = self.stage1_cfg["num_channels"][0]
        self.layer1 = self._make_layer(block, 64, c1, self.stage1_cfg["num_blocks"][0])
        pre = [c1 * block.expansion]
        for s in (2, 3, 4):
            cfg = extra["stage{}".format(s)]
            setattr(self, "stage{}_cfg".format(s), cfg)
            block = blocks_dict[cfg["block"]]
            widths = [c * block.expansion for c in cfg["num_channels"]]
            setattr(self, "transition{}".format(s - 1), self._make_transition_layer(pre, widths))
            stage, pre = self._make_stage(cfg, widths)
            setattr(self, "stage{}".format(s), stage)
        self._frozen_stages()

    def _make_transition_layer(self, pre, cur):
        layers = []
        for i in range(len(cur)):
            if i < len(pre):
                if cur[i] != pre[i]:
                    layers.append(nn.Sequential(nn.Conv2d(pre[i], cur[i], 3, 1, 1, bias=False),
                                                BatchNorm2d(cur[i], momentum=BN_MOMENTUM), nn.ReLU(inplace=True)))
                else:
                    layers.append(None)
            else:
                steps = i + 1 - len(pre)
                seq = []
                for j in range(steps):
                    co = cur[i] if j == steps - 1 else pre[-1]
                    seq.append(nn.Sequential(nn.Conv2d(pre[-1], co, 3, 2, 1, bias=False),
                                             BatchNorm2d(co, momentum=BN_MOMENTUM), nn.ReLU(inplace=True)))
                layers.append(nn.Sequential(*seq))
        return nn.ModuleList(layers)

    def _make_layer(self, block, inplanes, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                       BatchNorm2d(planes * block.expansion, momentum=BN_MOMENTUM))
        layers = [block(inplanes, planes, stride, downsample)]
        layers += [block(planes * block.expansion, planes) for _ in range(1, blocks)]
        return BlockChain(*layers)

    def _frozen_stages(self):
        if self.frozen_stages >= 0:
            for m in (self.conv1, self.bn1, self.conv2, self.bn2):
                for p in m.parameters():
                    p.requires_grad = False
        if self.frozen_stages == 1:
            for p in self.layer1.parameters():
                p.requires_grad = False

    def _make_stage(self, cfg, num_inchannels, multi_scale_output=True):
        block = blocks_dict[cfg["block"]]
        mods = []
        for i in range(cfg["num_modules"]):
            mso = multi_scale_output or i != cfg["num_modules"] - 1
            mods.append(HighResolutionModule(cfg["num_branches"], block, cfg["num_blocks"], num_inchannels,
                                             cfg["num_channels"], cfg["fuse_method"], mso))
            num_inchannels = mods[-1].get_num_inchannels()
        return nn.Sequential(*mods), num_inchannels

    def forward(self, x):
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
        if x.is_cuda and x.dtype == torch.float32 and not x.requires_grad and dt in (torch.float32, torch.bfloat16) \
                and x.shape[1] <= (8 if dt == torch.bfloat16 else 4):
            # activation dtype policy (bf16 activations, fp32 parameters / statistics), channels-last layout and the channel padding
            # of the first convolution in one launch
            x = nnf.image_to_channels_last(x, dt)
        else:
            x = x.to(dt).contiguous(memory_format=torch.channels_last)
        l1 = nnf.bwd_stats_link()                 # bn1's backward statistics ride on conv2's data-gradient launch
        x = nnf.conv_bn_act(x, self.conv1, self.bn1, nnf.ACT_RELU, stats_out=l1)
        x = nnf.conv_bn_act(x, self.conv2, self.bn2, nnf.ACT_RELU, stats_in=l1)
        x = self.layer1(x)
        ys = [x]
        for s in (2, 3, 4):
            trans = getattr(self, "transition{}".format(s - 1))
            nb = getattr(self, "stage{}_cfg".format(s))["num_branches"]
            xs = []
            # (transition1: layer1's 256-channel output feeds both new branches - its two data gradients share one buffer)
            # the last branch feeds the new branches' convolutions (accumulating) and, where it lives on, the stage itself: an alias
            keeps = len(ys) - 1 < nb and trans[len(ys) - 1] is None
            last, acc = nnf.fanout(ys[-1], sum(1 for i in range(nb) if trans[i] is not None), n_alias=2 if keeps else 1)
            if keeps:
                last, ident = last
                ys = list(ys[:-1]) + [ident]
            for i in range(nb):
                if trans[i] is None:
                    xs.append(ys[i])
                else:
                    xs.append(nnf.run_sequential(trans[i], last, grad_accum=acc))
            ys = getattr(self, "stage{}".format(s))(xs)
        return ys

    def train(self, mode=True):
        super().train(mode)
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self


def _factory(name):
    def build(pretrained=False, weight_path=None, norm_eval=False, frozen_stages=-1):
        model = HighResolutionNet(model_extra[name], norm_eval, zero_init_residual=False, frozen_stages=frozen_stages)
        if pretrained:
            if weight_path is None:
                raise FileNotFoundError("pretrained=True needs weight_path: this build has no network access "
                                        "(the reference downloads ImageNet HRNet weights, _hrnet_rssformer.py:32-37)")
            model.load_state_dict(torch.load(weight_path, map_location="cpu"), strict=False)
        return model
    build.__name__ = name
    return build


hrnetv2_w18 = _factory("hrnetv2_w18")
hrnetv2_w32 = _factory("hrnetv2_w32")
hrnetv2_w40 = _factory("hrnetv2_w40")
hrnetv2_w48 = _factory("hrnetv2_w48")
