"""Pipeline group (svo_pipeline_group_*): several stereo streams behind one caller thread, every stage one launch for the
lanes that reach it together, their bundle adjustments one device-resident solve launch.  Each lane must reproduce the
ORACLE pipeline of its own stream bit for bit (feature index sets, inlier counts, keyframe decisions, LM iteration counts
and poses) — i.e. exactly what a separate svo_pipeline gives (reference path: src/image_processor.cpp:18-163,
src/feature_tracker.cpp:18-67, src/bundle_adjuster.cpp:60-157, driver rule src/vo_node.cpp:141-148)."""
import numpy as np
import pytest

import oracle_lib as O
from test_pipeline import _seq, _ora_pipe

KEY = lambda r: (r.n_detected, r.n_tracked, r.n_inliers, r.n_new, r.is_keyframe, r.ba_iterations,
                 np.float32(r.av_parallax).view(np.uint32).item(), r.percent_lost if r.percent_lost == r.percent_lost else None, list(r.pose7))


def _group(S, ctx, p, maxc, md, mf, lanes, window=5):
    pp = S.pipeline_default_params()
    pp.cam.focal, pp.cam.cx, pp.cam.cy, pp.cam.baseline = p.focal, p.cx, p.cy, p.baseline
    pp.width, pp.height = p.width, p.height
    pp.max_corners, pp.min_feature_distance, pp.max_features, pp.window_size = maxc, md, mf, window
    pp.ba_max_time_s = 0.0  # deterministic: iteration cap only (SURVEY C-10)
    return S.PipelineGroup(ctx, pp, lanes)


@pytest.mark.gpu
@pytest.mark.parametrize("lanes,batch,md,maxc", [(1, 4, 30.0, 300), (4, 6, 10.0, 600), (7, 12, 12.0, 300), (20, 6, 14.0, 300)])
def test_hip_group_lanes_match_their_oracles(lanes, batch, md, maxc):
    """Lanes with DIFFERENT scenes (their keyframes fall on different frames: lanes sit in different stages of the chain at
    the same time and are batched in changing combinations); frames handed over in batches that cut the sequence."""
    import torch
    import stereo_vo_amd as S
    n = 12
    seqs = [_seq(n, seed=0x5EED0200 + 31 * i) for i in range(lanes)]
    # lane 1 (if any) moves differently: its keyframe cadence differs from the others'
    p0 = seqs[0][0]
    ctx = S.Context(p0.width, p0.height, max_batch=lanes * batch, max_corners=maxc, max_candidates=1 << 16, max_features=max(400, maxc))
    g = _group(S, ctx, p0, maxc, md, max(400, maxc), lanes)
    L = torch.from_numpy(np.stack([s[1] for s in seqs])).cuda()  # (lanes, n, H, W)
    R = torch.from_numpy(np.stack([s[2] for s in seqs])).cuda()
    got = [[] for _ in range(lanes)]
    for rep in range(2):  # the same frames twice from a reset group: also exercises reset
        g.reset()
        got = [[] for _ in range(lanes)]
        for b0 in range(0, n, batch):
            b = min(batch, n - b0)
            Lb, Rb = L[:, b0:b0 + b].contiguous(), R[:, b0:b0 + b].contiguous()
            res = g.process_batch_dev(Lb.data_ptr(), Rb.data_ptr(), b * p0.width * p0.height, b)
            torch.cuda.synchronize()
            for l in range(lanes):
                got[l] += res[l]
    stats = g.last_stats()
    for l in range(lanes):
        p, Lh, Rh = seqs[l]
        o = _ora_pipe(p, min_feature_distance=md, max_corners=maxc, max_features=max(400, maxc))
        ref = [o.process(Lh[k], Rh[k]) for k in range(n)]
        for k in range(n):
            assert KEY(got[l][k]) == KEY(ref[k]), (l, k, KEY(got[l][k]), KEY(ref[k]))
        ig, xg = g.get_tracked(l)
        io, xo = o.tracked()
        assert np.array_equal(ig, io) and np.array_equal(xg.view(np.uint32), xo.view(np.uint32)), l
        assert sum(r.is_keyframe for r in got[l]) >= 3
    if lanes > 1:  # stages are shared: fewer track launches than lane-stages they carried
        assert stats["track"][1] > stats["track"][0] > 0, stats
    g.close()
    ctx.close()


@pytest.mark.gpu
def test_hip_group_equals_separate_pipelines_on_identical_streams():
    """All lanes see the SAME stream: they move through the chain in lock step (every stage one launch for all lanes, one
    solve launch per keyframe) and every lane gives the single pipeline's results."""
    import torch
    import stereo_vo_amd as S
    n, lanes = 10, 5
    p, Lh, Rh = _seq(n, seed=0x5EED0300)
    ctx = S.Context(p.width, p.height, max_batch=lanes * n, max_corners=600, max_candidates=1 << 16, max_features=600)
    g = _group(S, ctx, p, 600, 10.0, 600, lanes)
    pp = g.prm
    single = S.Pipeline(ctx, pp)
    ref = single.process_batch(Lh, Rh)
    it, xt = single.tracked()
    L = torch.from_numpy(np.stack([Lh] * lanes)).cuda()
    R = torch.from_numpy(np.stack([Rh] * lanes)).cuda()
    res = g.process_batch_dev(L.data_ptr(), R.data_ptr(), n * p.width * p.height, n)
    torch.cuda.synchronize()
    for l in range(lanes):
        assert [KEY(r) for r in res[l]] == [KEY(r) for r in ref], l
        ig, xg = g.get_tracked(l)
        assert np.array_equal(ig, it) and np.array_equal(xg.view(np.uint32), xt.view(np.uint32))
    st = g.last_stats()
    n_kf = sum(r.is_keyframe for r in ref)
    # lanes in phase share launches (how many exactly depends on when completion words are seen: a lane never waits for another)
    assert st["bundle_adjust"][1] == n_kf * lanes and st["bundle_adjust"][0] < n_kf * lanes, st
    assert st["track"][1] == sum(1 for r in ref if r.n_tracked) * lanes and st["track"][0] < st["track"][1], st
    single.close()
    g.close()
    ctx.close()


@pytest.mark.gpu
def test_hip_group_tracks_across_batches_from_a_frame_that_was_not_the_batch_s_last():
    """The tracker reads level 0 of its pyramids from the caller's images in place; the image a lane will track FROM in the next
    batch is cloned (src/feature_tracker.cpp:14,66).  Usually that is the batch's last frame — here the last frames of the first
    batch are blank (fewer than 4 corners: the frame is skipped, SURVEY C-7), so the clone must be made of an earlier frame, and
    the caller then overwrites its buffer with the next batch."""
    import torch
    import stereo_vo_amd as S
    n, lanes, batch = 12, 3, 6
    seqs = [_seq(n, seed=0x5EED0400 + 17 * i) for i in range(lanes)]
    p0 = seqs[0][0]
    Ls = np.stack([s[1] for s in seqs]).copy()
    Rs = np.stack([s[2] for s in seqs]).copy()
    Ls[1, 4:6] = 128; Rs[1, 4:6] = 128  # lane 1: frames 4 and 5 (the end of batch 0) are blank
    Ls[2, 5:8] = 90; Rs[2, 5:8] = 90    # lane 2: blank across the batch boundary
    ctx = S.Context(p0.width, p0.height, max_batch=lanes * batch, max_corners=300, max_candidates=1 << 16, max_features=400)
    g = _group(S, ctx, p0, 300, 12.0, 400, lanes)
    buf_l = torch.empty((lanes, batch, p0.height, p0.width), dtype=torch.uint8, device="cuda")
    buf_r = torch.empty_like(buf_l)
    got = [[] for _ in range(lanes)]
    for b0 in range(0, n, batch):
        buf_l.copy_(torch.from_numpy(Ls[:, b0:b0 + batch]))  # the SAME device buffer for every batch
        buf_r.copy_(torch.from_numpy(Rs[:, b0:b0 + batch]))
        torch.cuda.synchronize()
        res = g.process_batch_dev(buf_l.data_ptr(), buf_r.data_ptr(), batch * p0.width * p0.height, batch)
        torch.cuda.synchronize()
        for l in range(lanes):
            got[l] += res[l]
    for l in range(lanes):
        o = _ora_pipe(seqs[l][0], min_feature_distance=12.0, max_corners=300, max_features=400)
        ref = [o.process(Ls[l, k], Rs[l, k]) for k in range(n)]
        for k in range(n):
            assert KEY(got[l][k]) == KEY(ref[k]), (l, k, KEY(got[l][k]), KEY(ref[k]))
        ig, xg = g.get_tracked(l)
        io, xo = o.tracked()
        assert np.array_equal(ig, io) and np.array_equal(xg.view(np.uint32), xo.view(np.uint32)), l
    assert got[1][4].n_detected < 4 and got[1][6].n_tracked > 0  # the blank frames were skipped, tracking went on from frame 3
    g.close()
    ctx.close()
