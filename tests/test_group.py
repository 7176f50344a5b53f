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
@pytest.mark.parametrize("lanes,batch,md,maxc", [(1, 4, 30.0, 300), (4, 6, 10.0, 600), (7, 12, 12.0, 300), (20, 6, 14.0, 300), (44, 4, 14.0, 300), (64, 3, 16.0, 300)])  # (round 5: up to 64 lanes per group)
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


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 3])
def test_hip_group_keeps_its_last_image_through_a_whole_blank_batch(batch):
    """A lane that neither tracks nor makes a keyframe during a WHOLE batch (every frame of it blank: fewer than 4 corners, C-7;
    with batch = 1 one blank frame is enough) still tracks its next frame from the image it tracked last: the reference holds a
    clone (src/feature_tracker.cpp:14,66).  The group's batch pyramids are double-buffered, so without a private copy that
    image would be overwritten by the batch after the blank one."""
    import torch
    import stereo_vo_amd as S
    n, lanes = 12, 3
    seqs = [_seq(n, seed=0x5EED0500 + 13 * i) for i in range(lanes)]
    p0 = seqs[0][0]
    Ls = np.stack([s[1] for s in seqs]).copy()
    Rs = np.stack([s[2] for s in seqs]).copy()
    b0_blank = 2 * batch  # the third batch of lane 1 is blank from its first to its last frame
    Ls[1, b0_blank:b0_blank + batch] = 77; Rs[1, b0_blank:b0_blank + batch] = 77
    Ls[2, 5:5 + 2 * batch] = 140; Rs[2, 5:5 + 2 * batch] = 140  # lane 2: blank over (at least) two consecutive batches
    ctx = S.Context(p0.width, p0.height, max_batch=lanes * batch, max_corners=300, max_candidates=1 << 16, max_features=400)
    g = _group(S, ctx, p0, 300, 12.0, 400, lanes)
    buf_l = torch.empty((lanes, batch, p0.height, p0.width), dtype=torch.uint8, device="cuda")
    buf_r = torch.empty_like(buf_l)
    got = [[] for _ in range(lanes)]
    for b0 in range(0, n - n % batch, batch):
        buf_l.copy_(torch.from_numpy(Ls[:, b0:b0 + batch]))
        buf_r.copy_(torch.from_numpy(Rs[:, b0:b0 + batch]))
        torch.cuda.synchronize()
        res = g.process_batch_dev(buf_l.data_ptr(), buf_r.data_ptr(), batch * p0.width * p0.height, batch)
        torch.cuda.synchronize()
        for l in range(lanes):
            got[l] += res[l]
    m = len(got[0])
    for l in range(lanes):
        o = _ora_pipe(seqs[l][0], min_feature_distance=12.0, max_corners=300, max_features=400)
        ref = [o.process(Ls[l, k], Rs[l, k]) for k in range(m)]
        for k in range(m):
            assert KEY(got[l][k]) == KEY(ref[k]), (l, k, KEY(got[l][k]), KEY(ref[k]))
    assert got[1][b0_blank].n_detected < 4 and any(r.n_tracked > 0 for r in got[1][b0_blank + batch:])
    g.close()
    ctx.close()


@pytest.mark.gpu
def test_hip_headline_load_24_lanes_full_size_two_groups_concurrently():
    """The configuration the bench's headline is quoted on, under its own load: a 24-lane group at 1241x376 / 1,500 corners /
    5-keyframe window / 16 frames, 5 repetitions from a reset group in one process WHILE a second group runs the same load on
    another thread (two groups = the bench's default shape: 48 lanes, the device-resident solves of both sharing the GPU).
    Every lane of the checked group must equal its oracle bit for bit in every repetition; the second group must reproduce
    itself.  (The cross-workgroup hand-overs of the solve kernel are exercised here at the load they are used at.)"""
    import threading
    import torch
    import stereo_vo_amd as S
    W, H, lanes, n, reps = 1241, 376, 24, 16, 5
    maxc, md, mf = 1500, 10.0, 2000
    p = S.synth_default(W, H)
    seeds = [0x5EED0001 + i for i in range(2 * lanes)]  # the bench's seeds

    def render(seed):
        q = S.synth_default(W, H)
        q.seed = seed
        fr = [S.synth_render(q, i) for i in range(n)]
        return q, np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr])

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=8) as ex:
        data = list(ex.map(render, seeds))
    groups = []
    for gi in range(2):
        ctx = S.Context(W, H, max_batch=lanes * n, max_corners=maxc, max_candidates=1 << 16, max_features=mf)
        pp = S.pipeline_default_params()
        pp.cam.focal, pp.cam.cx, pp.cam.cy, pp.cam.baseline = p.focal, p.cx, p.cy, p.baseline
        pp.width, pp.height = W, H
        pp.max_corners, pp.quality, pp.min_feature_distance, pp.max_features, pp.window_size = maxc, 0.02, md, mf, 5
        pp.ba_max_time_s = 0.0
        g = S.PipelineGroup(ctx, pp, lanes)
        mine = data[gi::2]  # dealt round-robin as bench.py does
        L = torch.from_numpy(np.stack([d[1] for d in mine])).cuda()
        R = torch.from_numpy(np.stack([d[2] for d in mine])).cuda()
        groups.append(dict(ctx=ctx, g=g, L=L, R=R, data=mine, out=[], err=None))
    torch.cuda.synchronize()

    def work(G):
        try:
            for _ in range(reps):
                G["g"].reset()
                G["out"].append(G["g"].process_batch_dev(G["L"].data_ptr(), G["R"].data_ptr(), n * W * H, n))
        except Exception as e:  # noqa: BLE001
            G["err"] = e

    th = [threading.Thread(target=work, args=(G,)) for G in groups]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    for G in groups:
        assert G["err"] is None, G["err"]
        assert len(G["out"]) == reps
    # group 0: every lane against its oracle, every repetition
    G = groups[0]

    def oracle_lane(d):
        q, Lh, Rh = d
        o = O.Pipeline(focal=q.focal, cx=q.cx, cy=q.cy, baseline=q.baseline, width=W, height=H, max_corners=maxc, quality=0.02,
                       min_feature_distance=md, parallax_thresh=20.0, window_size=5, max_features=mf, ba_max_iterations=50, num_threads=2)
        return [KEY(o.process(Lh[k], Rh[k])) for k in range(n)]

    with ThreadPoolExecutor(max_workers=8) as ex:
        refs = list(ex.map(oracle_lane, G["data"]))
    for rep in range(reps):
        for l in range(lanes):
            got = [KEY(r) for r in G["out"][rep][l]]
            assert got == refs[l], (rep, l, [k for k in range(n) if got[k] != refs[l][k]][:3])
    # group 1: identical frames every repetition => identical bits
    G1 = groups[1]
    for rep in range(1, reps):
        for l in range(lanes):
            assert [KEY(r) for r in G1["out"][rep][l]] == [KEY(r) for r in G1["out"][0][l]], (rep, l)
    for G in groups:
        G["g"].close()
        G["ctx"].close()


@pytest.mark.gpu
def test_hip_group_streaming_entry_equals_the_device_pointer_entry():
    """The host-pointer entries (pinned staging slots, upload of batch b+1 overlapped with the processing of batch b; and the
    copying convenience call) give every lane the results of svo_pipeline_group_process_batch_dev, bit for bit
    (reference: the images are host cv::Mat copies handed over frame by frame, src/vo_node.cpp:70-73,141-143)."""
    import torch
    import stereo_vo_amd as S
    n, lanes, batch = 12, 4, 4
    seqs = [_seq(n, seed=0x5EED0600 + 7 * i) for i in range(lanes)]
    p0 = seqs[0][0]
    Ls = np.stack([s[1] for s in seqs])
    Rs = np.stack([s[2] for s in seqs])
    ctx = S.Context(p0.width, p0.height, max_batch=lanes * batch, max_corners=300, max_candidates=1 << 16, max_features=400)
    g = _group(S, ctx, p0, 300, 12.0, 400, lanes)
    # (1) device-pointer entry
    want = [[] for _ in range(lanes)]
    for b0 in range(0, n, batch):
        dl, dr = torch.from_numpy(Ls[:, b0:b0 + batch].copy()).cuda(), torch.from_numpy(Rs[:, b0:b0 + batch].copy()).cuda()
        res = g.process_batch_dev(dl.data_ptr(), dr.data_ptr(), batch * p0.width * p0.height, batch)
        torch.cuda.synchronize()
        for l in range(lanes):
            want[l] += res[l]
    # (2) streaming: fill slot, upload, process the previous one
    g.reset()
    got = [[] for _ in range(lanes)]
    nb = n // batch
    sl, sr = g.staging(0)
    sl[:, :batch], sr[:, :batch] = Ls[:, 0:batch], Rs[:, 0:batch]
    g.upload(0, batch)
    for b in range(nb):
        if b + 1 < nb:
            nl, nr = g.staging((b + 1) & 1)
            nl[:, :batch], nr[:, :batch] = Ls[:, (b + 1) * batch:(b + 2) * batch], Rs[:, (b + 1) * batch:(b + 2) * batch]
            g.upload((b + 1) & 1, batch)
        res = g.process_uploaded(b & 1, batch)
        for l in range(lanes):
            got[l] += res[l]
    for l in range(lanes):
        assert [KEY(r) for r in got[l]] == [KEY(r) for r in want[l]], l
    # (3) the copying convenience entry, with a batch shorter than the slots (strided upload)
    g.reset()
    got = [[] for _ in range(lanes)]
    for b0 in range(0, n, 3):
        res = g.process_batch(Ls[:, b0:b0 + 3], Rs[:, b0:b0 + 3])
        for l in range(lanes):
            got[l] += res[l]
    g.reset()
    want3 = [[] for _ in range(lanes)]
    for b0 in range(0, n, 3):
        dl, dr = torch.from_numpy(Ls[:, b0:b0 + 3].copy()).cuda(), torch.from_numpy(Rs[:, b0:b0 + 3].copy()).cuda()
        res = g.process_batch_dev(dl.data_ptr(), dr.data_ptr(), 3 * p0.width * p0.height, 3)
        torch.cuda.synchronize()
        for l in range(lanes):
            want3[l] += res[l]
    for l in range(lanes):
        assert [KEY(r) for r in got[l]] == [KEY(r) for r in want3[l]], l
    g.close()
    ctx.close()


@pytest.mark.gpu
def test_hip_group_landmark_store_too_small_fails_loudly():
    """The PnP launch reads world points from the lane's device-resident landmark store, keyed by feature id modulo the
    capacity.  A capacity below the span of live ids (forced here through the test hook SVO_GROUP_STORE_KEYFRAMES=1: room for
    ONE keyframe's ids) lets a new landmark overwrite a live one's entry; the launch must report the entry found under a foreign
    id — an error, never a silently wrong pose."""
    import os
    import torch
    import stereo_vo_amd as S
    n = 40
    p0, L, R = _seq(n, seed=0x5EED0901)[:3]
    ctx = S.Context(p0.width, p0.height, max_batch=8, max_corners=300, max_candidates=1 << 16, max_features=400)
    os.environ["SVO_GROUP_STORE_KEYFRAMES"] = "1"
    try:
        g = _group(S, ctx, p0, 300, 12.0, 400, 1)
    finally:
        del os.environ["SVO_GROUP_STORE_KEYFRAMES"]
    def run(b0):
        dl, dr = torch.from_numpy(L[None, b0:b0 + 8].copy()).cuda(), torch.from_numpy(R[None, b0:b0 + 8].copy()).cuda()
        out = g.process_batch_dev(dl.data_ptr(), dr.data_ptr(), 8 * p0.width * p0.height, 8)
        torch.cuda.synchronize()
        return [bytes(r) for r in out[0]]
    good, failed_at = [], None
    with pytest.raises(S.api.SvoError, match="landmark store"):
        for b0 in range(0, n, 8):
            failed_at = b0
            good.append(run(b0))
    assert failed_at is not None and failed_at >= 8, "the first batch fits even one keyframe's ids"
    # the documented recovery: after the error a reset gives a working group again — the report word is cleared and
    # the store refilled, so the batches that succeeded before succeed again with the same bits
    g.reset()
    for i, b0 in enumerate(range(0, failed_at, 8)):
        assert run(b0) == good[i], b0
    g.close()
    ctx.close()


@pytest.mark.gpu
def test_hip_group_host_driven_solves_fill_the_landmark_store_too():
    """A window that is not eligible (or not admitted) for the device-resident solve is solved by the host-driven loop on a
    worker; its landmarks then reach the lane's landmark store through one scatter launch.  Forced for every window through the
    test hook SVO_GROUP_HOST_SOLVES=1 (read when the library first launches solves, hence a child process): every lane must
    still be its oracle, bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import stereo_vo_amd as S
from test_pipeline import _seq, _ora_pipe
from test_group import _group, KEY
n, lanes, batch = 16, 3, 8
seqs = [_seq(n, seed=0x5EED0A00 + 11 * i) for i in range(lanes)]
p0 = seqs[0][0]
Ls = np.stack([s[1] for s in seqs]); Rs = np.stack([s[2] for s in seqs])
ctx = S.Context(p0.width, p0.height, max_batch=lanes * batch, max_corners=300, max_candidates=1 << 16, max_features=400)
g = _group(S, ctx, p0, 300, 12.0, 400, lanes)
got = [[] for _ in range(lanes)]
for b0 in range(0, n, batch):
    dl, dr = torch.from_numpy(Ls[:, b0:b0 + batch].copy()).cuda(), torch.from_numpy(Rs[:, b0:b0 + batch].copy()).cuda()
    res = g.process_batch_dev(dl.data_ptr(), dr.data_ptr(), batch * p0.width * p0.height, batch)
    torch.cuda.synchronize()
    for l in range(lanes):
        got[l] += res[l]
for l in range(lanes):
    p, Lh, Rh = seqs[l]
    o = _ora_pipe(p, min_feature_distance=12.0, max_corners=300, max_features=400)
    ref = [o.process(Lh[k], Rh[k]) for k in range(n)]
    for k in range(n):
        assert KEY(got[l][k]) == KEY(ref[k]), (l, k, KEY(got[l][k]), KEY(ref[k]))
assert sum(q.is_keyframe for q in got[0]) >= 3
print("host-driven solves ok", sum(q.is_keyframe for q in got[0]))
''' % (root, os.path.join(root, "tests"))
    e = dict(os.environ)
    e["SVO_GROUP_HOST_SOLVES"] = "1"
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "host-driven solves ok" in out.stdout, out.stderr[-3000:]


_CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np, torch
import stereo_vo_amd as S
from test_pipeline import _seq, _ora_pipe
from test_group import _group, KEY
n, lanes, batch, seed0 = %(n)d, %(lanes)d, %(batch)d, %(seed)d
seqs = [_seq(n, seed=seed0 + 11 * i) for i in range(lanes)]
p0 = seqs[0][0]
Ls = np.stack([s[1] for s in seqs]); Rs = np.stack([s[2] for s in seqs])
ctx = S.Context(p0.width, p0.height, max_batch=lanes * batch, max_corners=300, max_candidates=1 << 16, max_features=400)
g = _group(S, ctx, p0, 300, 12.0, 400, lanes)
for rep in range(%(reps)d):
    g.reset()
    got = [[] for _ in range(lanes)]
    for b0 in range(0, n, batch):
        dl, dr = torch.from_numpy(Ls[:, b0:b0 + batch].copy()).cuda(), torch.from_numpy(Rs[:, b0:b0 + batch].copy()).cuda()
        res = g.process_batch_dev(dl.data_ptr(), dr.data_ptr(), batch * p0.width * p0.height, batch)
        torch.cuda.synchronize()
        for l in range(lanes):
            got[l] += res[l]
    for l in range(lanes):
        p, Lh, Rh = seqs[l]
        o = _ora_pipe(p, min_feature_distance=12.0, max_corners=300, max_features=400)
        ref = [o.process(Lh[k], Rh[k]) for k in range(n)]
        for k in range(n):
            assert KEY(got[l][k]) == KEY(ref[k]), (rep, l, k, KEY(got[l][k]), KEY(ref[k]))
print("child ok", sum(q.is_keyframe for q in got[0]), flush=True)
'''


def _child_code(**kw):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return _CHILD % dict(root=root, tests=os.path.join(root, "tests"), **kw)


@pytest.mark.gpu
def test_hip_group_solves_that_give_up_are_rerun_and_lose_no_frame():
    """a device-resident solve of the wide form (ba_lm_kernel: workgroups that wait for each other) may fail to become
    co-resident and give up within its bound.  The frame must not be lost: the problem is still loaded, the library re-runs it
    (compact form — one workgroup, nothing to wait for — else the host-driven loop) and the lane goes on.  Forced here through the
    test hook SVO_BA_TEST_GIVEUP=3 (every third wide launch of an adjuster reports "gave up" at once; read once per process,
    hence a child process): every lane of the group must still be its oracle, bit for bit."""
    import os
    import subprocess
    import sys
    e = dict(os.environ)
    e["SVO_BA_TEST_GIVEUP"] = "3"
    out = subprocess.run([sys.executable, "-c", _child_code(n=16, lanes=5, batch=8, seed=0x5EED0B00, reps=1)], env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "child ok" in out.stdout, out.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"SVO_GROUP_CHAIN_FUSED": "0"}, {"SVO_GROUP_GATHER_US": "200"},
                                 {"SVO_GROUP_COMPACT_LINES": "1", "SVO_BA_BUDGET_PERCENT": "25"},
                                 {"SVO_GROUP_COMPACT_LINES": "2", "SVO_GROUP_BA_LINES": "1", "SVO_BA_BUDGET_PERCENT": "12"},
                                 ])
def test_hip_group_optional_paths_keep_parity(env):
    """Round 5's scheduling knobs of a pipeline group must not change a bit: the keyframe chain with a host turn between the PnP and the
    stereo launch again (default: the stereo launch rides right behind PnP and reads the reprojection matrix from the device record,
    host/chain_math.h), the gather policy, compact lines (the admission budget cut so far that most solves take them).  Read once per process: one child each; 6 lanes x 16 frames against the oracle."""
    import os
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", _child_code(n=16, lanes=6, batch=8, seed=0x5EED0D00, reps=2)], env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "child ok" in out.stdout, out.stderr[-3000:]


@pytest.mark.gpu
def test_hip_two_processes_share_one_gpu_without_losing_a_solve():
    """Two PROCESSES drive pipeline groups on the same GPU at once.  The admission of the wide solves counts the workgroups of every live
    process on the device (/dev/shm/svo_admit_<PCI bus id>, csrc/ba.hip), here with 150 % of the budget to make refusals and
    crowded launches common; whatever is refused or gives up takes the compact form or the host-driven loop.  Both processes must
    finish with every lane equal to its oracle (the reference never loses a keyframe to scheduling, src/bundle_adjuster.cpp:137-157)."""
    import os
    import subprocess
    import sys
    e = dict(os.environ)
    e["SVO_BA_BUDGET_PERCENT"] = "150"
    procs = [subprocess.Popen([sys.executable, "-c", _child_code(n=16, lanes=10, batch=8, seed=0x5EED0C00 + 1000 * i, reps=3)], env=e,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(2)]
    outs = [p.communicate(timeout=500) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "child ok" in so, se[-3000:]
    shm = [f for f in os.listdir("/dev/shm") if f.startswith("svo_admit_")]
    assert shm, "the cross-process admission table was never created"
