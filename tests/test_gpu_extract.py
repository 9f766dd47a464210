"""GPU parity: HIP extraction (through the C ABI) vs the CPU oracle -- bit-exact keypoints and descriptors,
stage by stage (pyramid, FAST candidates, blur) and end to end, plus golden fixtures and edge cases."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["separate blur kernels", "blur fused into k_describe", "fused + early FAST of levels 0-1",
                                      "fused + resize chain (FAST cells do not emit the next level)"])
def pipeline_mode(request, opts):
    """Every test of this file runs with every pipeline variant (the options DCS_ORB_FUSED_BLUR / DCS_ORB_FAST_SPLIT / DCS_ORB_EMIT are copied when an extractor
    handle is created): the library picks the blur variant per call from the pyramid pixels per feature and the batch size, so small
    test batches would only ever see the fused one; the early FAST launch (two k_fast_cells launches on two streams) is opt-in."""
    opts("DCS_ORB_FUSED_BLUR", 0 if request.param.startswith("separate") else 1)
    opts("DCS_ORB_FAST_SPLIT", int("2" if "early" in request.param else "0"))
    # round 5: by default the FAST cells of level l write level l + 1 (k_fast_cells<EMIT>) -- whenever the blur is fused and no early FAST launch
    # is asked for, i.e. in the second variant; the fourth keeps the fused describe on the round-4 resize chain
    opts("DCS_ORB_EMIT", int("0" if "resize chain" in request.param else "15"))     # n > 0: the cells of levels [0, n) emit, whatever the batch size


GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _same(kp_a, desc_a, kp_b, desc_b):
    assert len(kp_a) == len(kp_b)
    assert kp_a.tobytes() == kp_b.tobytes()
    assert np.array_equal(desc_a, desc_b)


@pytest.fixture
def ext1000(pkg):
    e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=4)
    yield e
    e.close()


def test_tables_match_oracle(pkg, oracle, ext1000):
    t, o = ext1000.tables(), oracle.OrbOracle(1000, 1.2, 8, 20, 7).tables()
    for k in ("scale", "inv_scale", "sigma2", "inv_sigma2", "n_per_level"):
        assert np.array_equal(t[k], o[k]), k


def test_stages_bit_exact_640x480(pkg, oracle, synth, ext1000):
    img0, img1 = synth.frame_pair(640, 480, 0, 0)
    kps, descs = ext1000.extract_batch([img0, img1])
    for i, img in enumerate((img0, img1)):
        o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
        okp, odesc = o.extract(img)
        for l in range(8):
            assert ext1000.level_dims(l) == o.level_dims(l)
            assert np.array_equal(ext1000.level_image(i, l), o.level_image(l)), ("pyramid", i, l)
            assert np.array_equal(ext1000.level_image(i, l, blurred=True), oracle.gauss7_u8(o.level_image(l))), ("blur", i, l)
            assert ext1000.level_candidates(i, l).tobytes() == o.level_candidates(l).tobytes(), ("fast", i, l)
        _same(kps[i], descs[i], okp, odesc)


@pytest.mark.parametrize("w,h,n", [(640, 480, 1300), (1280, 720, 2000), (752, 480, 500), (333, 245, 300)])
def test_end_to_end_bit_exact(pkg, oracle, synth, w, h, n):
    imgs = synth.frame_pair(1280, 720, 1, 2) if (w, h) == (1280, 720) else synth.frame_pair(640, 480, 2, 1)
    imgs = [np.ascontiguousarray(im[:h, :w]) if im.shape != (h, w) else im for im in imgs]
    if imgs[0].shape != (h, w):                      # 752x480: pad from the big scene
        big = synth.frame_pair(1280, 720, 1, 3)
        imgs = [np.ascontiguousarray(b[:h, :w]) for b in big]
    e = pkg.ORBextractor(n, 1.2, 8, 20, 7, max_images=2)
    kps, descs = e.extract_batch(imgs, cap=n + 200)
    for i in range(2):
        okp, odesc = oracle.OrbOracle(n, 1.2, 8, 20, 7).extract(imgs[i], cap=n + 200)
        _same(kps[i], descs[i], okp, odesc)
    e.close()


@pytest.mark.parametrize("w,h", [(333, 245), (642, 479), (701, 350), (1277, 719)])
def test_emitting_fast_equals_the_resize_chain_on_odd_sizes(pkg, opts, synth, w, h):
    """The default large-batch path -- the FAST cells of level l write level l + 1 (k_fast_cells<EMIT>: buffer-resource stores, per-cell tap windows,
    the frame no cell's ROI reaches as extra workgroups) -- is chosen automatically only from 2.5e7 level-0 pixels per call, which no test batch
    reaches: here it is FORCED (DCS_ORB_EMIT = all levels) on image sizes that are not multiples of 4 / 30 and held against the resize chain
    (DCS_ORB_EMIT = 0) of the same images: every pyramid level, key point and descriptor bit for bit. emit_levels() > 0 is asserted so that a
    silent fallback to the chain cannot make the comparison vacuous."""
    big = synth.frame_pair(1280, 720, 1, 5)
    imgs = [np.ascontiguousarray(b[:h, :w]) for b in big]
    opts("DCS_ORB_FUSED_BLUR", 1); opts("DCS_ORB_FAST_SPLIT", 0)
    opts("DCS_ORB_EMIT", 15)
    e1 = pkg.ORBextractor(800, 1.2, 8, 20, 7, max_images=2)
    k1, d1 = e1.extract_batch(imgs, cap=1000)
    assert e1.emit_levels() > 0, "the emitting FAST did not run"
    lv1 = [[e1.level_image(i, l) for l in range(8)] for i in range(2)]
    opts("DCS_ORB_EMIT", 0)
    e0 = pkg.ORBextractor(800, 1.2, 8, 20, 7, max_images=2)
    k0, d0 = e0.extract_batch(imgs, cap=1000)
    assert e0.emit_levels() == 0
    for i in range(2):
        for l in range(8):
            assert np.array_equal(lv1[i][l], e0.level_image(i, l)), ("level", i, l)
        _same(k1[i], d1[i], k0[i], d0[i])
    e1.close(); e0.close()


@pytest.mark.parametrize("host_threads", [1, 4])
def test_host_quadtree_mode_matches_too(pkg, oracle, synth, host_threads):
    """host_threads > 0 selects the host-side sort/scan quadtree (csrc/octree.cpp) instead of k_octree."""
    imgs = list(synth.frame_pair(640, 480, 4, 0)) + [np.random.default_rng(1).integers(0, 256, (480, 640), dtype=np.uint8)]
    e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=3, host_threads=host_threads)
    kps, descs = e.extract_batch(imgs)
    for i in range(3):
        okp, odesc = oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(imgs[i])
        _same(kps[i], descs[i], okp, odesc)
    e.close()


def test_single_image_operator_and_strided_input(pkg, oracle, synth, ext1000):
    img, _ = synth.frame_pair(640, 480, 3, 0)
    kp, desc = ext1000(img)
    okp, odesc = oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(img)
    _same(kp, desc, okp, odesc)
    # non-default FAST thresholds / levels / scale
    e = pkg.ORBextractor(600, 1.5, 5, 30, 10, max_images=1)
    kp, desc = e(img)
    okp, odesc = oracle.OrbOracle(600, 1.5, 5, 30, 10).extract(img)
    _same(kp, desc, okp, odesc)
    e.close()


def test_edge_cases(pkg, oracle, synth, ext1000):
    # textureless image -> zero keypoints, no error (ORBextractor.cc:1064-1065)
    kp, desc = ext1000(np.full((480, 640), 128, np.uint8))
    assert len(kp) == 0 and desc.shape == (0, 32)
    # saturated / extreme-contrast checkerboard: many equal scores (NMS ties) and saturating blur
    yy, xx = np.mgrid[0:480, 0:640]
    chk = (((xx // 16) + (yy // 16)) % 2 * 255).astype(np.uint8)
    kp, desc = ext1000(chk)
    okp, odesc = oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(chk)
    _same(kp, desc, okp, odesc)
    # pure noise: every cell full of candidates (stress for per-cell slot capacity)
    noise = np.random.default_rng(0).integers(0, 256, (480, 640), dtype=np.uint8)
    kp, desc = ext1000(noise)
    okp, odesc = oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(noise)
    _same(kp, desc, okp, odesc)
    # capacity error is reported, not silently truncated
    img, _ = synth.frame_pair(640, 480, 0, 0)
    with pytest.raises(pkg.DcsError) as ei:
        ext1000.extract_batch([img], cap=100)
    assert ei.value.rc == pkg.abi.DCS_ERR_CAPACITY
    # too many images for the handle
    with pytest.raises(pkg.DcsError):
        ext1000.extract_batch([img] * 5)
    # image smaller than the border allows
    with pytest.raises(pkg.DcsError):
        ext1000(np.zeros((30, 30), np.uint8))


def test_golden_fixture_and_hashes(pkg, synth):
    g = np.load(os.path.join(GOLDEN, "extract_320x240_n300.npz"))
    e = pkg.ORBextractor(300, 1.2, 8, 20, 7, max_images=1)
    kp, desc = e(g["image"])
    assert kp.tobytes() == g["keypoints"].tobytes() and np.array_equal(desc, g["descriptors"])
    e.close()
    gold = json.load(open(os.path.join(GOLDEN, "extract_hashes.json")))
    for key, rec in gold.items():
        img = synth.frame_pair(rec["width"], rec["height"], rec["stream"], rec["frame"])[rec["cam"]]
        if hashlib.sha256(img.tobytes()).hexdigest() != rec["image_sha256"]:
            pytest.skip("synthetic generator differs on this numpy build")
        e = pkg.ORBextractor(rec["nfeatures"], 1.2, 8, 20, 7, max_images=1)
        kp, desc = e(img, cap=rec["nfeatures"] + 200)
        assert len(kp) == rec["n_keypoints"], key
        assert hashlib.sha256(kp.tobytes()).hexdigest() == rec["keypoints_sha256"], key
        assert hashlib.sha256(desc.tobytes()).hexdigest() == rec["descriptors_sha256"], key
        e.close()


def test_device_resident_batch_matches_host_api(pkg, oracle, synth):
    import torch
    B = 6
    imgs = []
    for f in range(B // 2):
        imgs.extend(synth.frame_pair(640, 480, 0, f))
    e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=B)
    cap = e.default_cap()
    d_img = torch.from_numpy(np.stack(imgs)).cuda()
    d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    e.extract_batch_device(d_img, d_kp, d_desc, d_n, cap, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    n = d_n.cpu().numpy()
    kp_all = d_kp.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
    desc_all = d_desc.cpu().numpy()
    for i in range(B):
        okp, odesc = oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(imgs[i])
        assert n[i] == len(okp)
        assert kp_all[i, :n[i]].tobytes() == okp.tobytes()
        assert np.array_equal(desc_all[i, :n[i]], odesc)
    t = e.last_timing()
    assert t["total_us"] > 0 and t["fast_us"] > 0 and t["describe_us"] > 0
    # dcs_orb_set_timing: 1 = only the FAST stage is bracketed (same results, the other entries read 0), 0 = no markers
    ref = (n.copy(), kp_all.copy(), desc_all.copy())
    for mode in (1, 0, 2):
        e.set_timing(mode)
        d_kp.zero_(); d_desc.zero_(); d_n.zero_()
        e.extract_batch_device(d_img, d_kp, d_desc, d_n, cap, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(d_n.cpu().numpy(), ref[0]) and np.array_equal(d_desc.cpu().numpy(), ref[2])
        assert d_kp.cpu().numpy().view(np.uint8).reshape(B, cap, 28).tobytes() == ref[1].tobytes()
        if mode == 0:
            with pytest.raises(Exception):
                e.last_timing()
        else:
            t = e.last_timing()
            assert t["fast_us"] > 0 and (t["describe_us"] > 0) == (mode == 2) and (t["total_us"] > 0) == (mode == 2)
    with pytest.raises(Exception):
        e.set_timing(3)
    e.close()


def test_deep_quadtree_leaves_the_histogram_fast_path(pkg, oracle, synth):
    """Corner patches on a geometric diagonal: one node splits per level, so the quota drives the quadtree deeper
    than the 6-level LDS histogram pyramid and the sort-based kernel redoes those (image, level) tasks. The usual
    scene stays on the fast path. Both are bit-exact."""
    rng = np.random.default_rng(0)
    img = np.full((480, 640), 120, np.uint8)
    for k in range(1, 10):
        x, y, s = int(30 + 560 * (1 - 2.0 ** -k)), int(30 + 400 * (1 - 2.0 ** -k)), max(4, 12 - k // 2)
        img[y:y + s, x:x + s] = rng.integers(0, 256, (s, s), dtype=np.uint8)
    e = pkg.ORBextractor(1000, 1.2, 8, 12, 5, max_images=1)
    kp, desc = e(img)
    assert e.quadtree_fallbacks() >= 3
    okp, odesc = oracle.OrbOracle(1000, 1.2, 8, 12, 5).extract(img)
    assert len(okp) > 100
    _same(kp, desc, okp, odesc)
    e.close()
    e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=1)
    e(synth.frame_pair(640, 480, 0, 0)[0])
    assert e.quadtree_fallbacks() == 0
    e.close()


def test_wide_image_with_small_quota(pkg, oracle, synth):
    """Wide, short images start the quadtree from round(w/h) >= 3 initial nodes and the reference splits all of them once
    before it looks at the quota (ORBextractor.cc:594-673): with a small per-level quota a level keeps up to 4 * nIni
    keypoints, more than N + 3. Found by scratch/stress_parity.py; the output slots are sized for it now."""
    src = synth.frame_pair(1280, 720, 0, 0)[0]
    exceeded = 0
    for (w, h, nf, nl, sf, ini, mn) in [(836, 144, 68, 4, 1.27, 21, 17), (706, 199, 62, 9, 1.129, 37, 26), (650, 253, 73, 9, 1.174, 33, 20),
                                        (757, 176, 258, 9, 1.1255, 29, 26)]:
        img = np.ascontiguousarray(src[100:100 + h, 200:200 + w])
        e = pkg.ORBextractor(nf, sf, nl, ini, mn, max_images=1)
        need = e.required_cap(h, w)
        assert need > nf + 8 * nl
        with pytest.raises(pkg.DcsError):
            e.extract_batch([img], cap=nf + 8 * nl)
        kp, desc = e(img)
        okp, odesc = oracle.OrbOracle(nf, sf, nl, ini, mn).extract(img, cap=need)
        exceeded += len(okp) > nf + 8 * nl
        _same(kp, desc, okp, odesc)
        e.close()
    assert exceeded >= 1


@pytest.mark.parametrize("lead,stride", [(1, 323), (4, 324), (16, 336), (0, 320)])
def test_device_api_unaligned_pointer_and_stride(pkg, oracle, synth, lead, stride):
    """HBM-resident input at every alignment class of the level-0 loads: neither 4-byte aligned nor 4-byte strided (byte-wise
    fallbacks), 4-byte but not 16-byte aligned (dword path of IC_Angle), 16-byte aligned with padding and with a stride that
    leaves only 3 bytes behind the last column (the 16-byte chunks of IC_Angle must stay inside the row)."""
    import torch
    B, rows, cols = 2, 240, 317
    base = synth.frame_pair(640, 480, 1, 0)
    imgs = [np.ascontiguousarray(b[100:100 + rows, 50:50 + cols]) for b in base]
    buf = torch.zeros(B * rows * stride + 64, dtype=torch.uint8, device="cuda")
    assert buf.data_ptr() % 16 == 0
    view = buf[lead:lead + B * rows * stride].view(B, rows, stride)
    view[:, :, :cols] = torch.from_numpy(np.stack(imgs)).cuda()
    view[:, :, cols:] = 255                                            # padding must never be read as image
    e = pkg.ORBextractor(400, 1.2, 8, 20, 7, max_images=B)
    cap = e.default_cap()
    d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    assert view.data_ptr() % 16 == lead % 16
    e.extract_batch_device(view, d_kp, d_desc, d_n, cap, stream=torch.cuda.current_stream().cuda_stream, cols=cols)
    torch.cuda.synchronize()
    n = d_n.cpu().numpy()
    kp_all = d_kp.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
    for i in range(B):
        okp, odesc = oracle.OrbOracle(400, 1.2, 8, 20, 7).extract(imgs[i])
        assert n[i] == len(okp) and len(okp) > 50
        assert kp_all[i, :n[i]].tobytes() == okp.tobytes()
        assert np.array_equal(d_desc[i, :n[i]].cpu().numpy(), odesc)
    e.close()


def test_general_quadtree_kernel_alone(pkg, oracle, synth, opts):
    """option DCS_OCTREE_FORCE_GENERAL = 1 sends every (image, level) task to the sort-based kernel (read per call); the result must
    equal the oracle's as well."""
    opts("DCS_OCTREE_FORCE_GENERAL", 1)
    imgs = list(synth.frame_pair(640, 480, 4, 1))
    e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=2)
    kps, descs = e.extract_batch(imgs)
    assert e.quadtree_fallbacks() == 16
    for i in range(2):
        okp, od = oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(imgs[i])
        assert kps[i].tobytes() == okp.tobytes() and np.array_equal(descs[i], od)
    e.close()


def test_candidate_buffer_overflow_is_reported_in_band(pkg, synth, opts):
    """The asynchronous device API cannot return an error after the fact: when the FAST candidates of a batch exceed the
    handle's dense buffer (provoked here with the option DCS_ORB_DENSE_CAP, copied when the handle is created)
    every count of the call is DCS_ERR_CAPACITY (-2) instead of a number of keypoints, and the host-buffer API fails."""
    import torch
    opts("DCS_ORB_DENSE_CAP", 3000)
    imgs = list(synth.frame_pair(640, 480, 4, 1))
    e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=2)
    cap = e.default_cap()
    d_img = torch.from_numpy(np.stack(imgs)).cuda()
    d_kp = torch.zeros((2, cap, 7), dtype=torch.float32, device="cuda"); d_desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.full((2,), 7, dtype=torch.int32, device="cuda")
    e.extract_batch_device(d_img, d_kp, d_desc, d_n, cap, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert d_n.tolist() == [-2, -2], d_n.tolist()
    with pytest.raises(pkg.DcsError) as ei:
        e.extract_batch(imgs)
    assert ei.value.rc == -2
    e.close()


def test_gpu_sincosf_equals_libm(pkg, oracle):
    """The describe kernel's steering coefficients: glibc's cosf / sinf evaluated on the GPU, bit for bit the host libm's
    (the functions the reference calls, ORBextractor.cc:112-113), on 4 M angles of [0, 2 pi] and the quadrant boundaries."""
    hi = np.float32(6.2832).view(np.uint32)
    x = np.arange(0, int(hi), 263, dtype=np.uint32).view(np.float32)
    x = np.concatenate([x, (np.linspace(0, 360, 100001, dtype=np.float32) * np.float32(np.float32(3.1415926535897932384626433832795) / np.float32(180.0)))])
    gc, gs = pkg.abi.debug_sincosf(x)
    hc, hs = oracle.sincosf(x)
    assert np.array_equal(gc.view(np.uint32), hc.view(np.uint32)) and np.array_equal(gs.view(np.uint32), hs.view(np.uint32))


def test_fused_blur_describe_keeps_the_blurred_debug_level(pkg, oracle, synth, opts):
    """With the fused describe no blur kernel runs in the product path: the blurred debug level is made on demand and is still the oracle's."""
    opts("DCS_ORB_FUSED_BLUR", int("1"))
    imgs = list(synth.frame_pair(640, 480, 2, 1)) + [np.random.default_rng(5).integers(0, 256, (480, 640), dtype=np.uint8)]
    e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=3)
    kps, descs = e.extract_batch(imgs)
    for i in range(3):
        o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
        okp, odesc = o.extract(imgs[i])
        _same(kps[i], descs[i], okp, odesc)
        assert np.array_equal(e.level_image(i, 2, blurred=True), oracle.gauss7_u8(o.level_image(2)))
    e.close()


@pytest.mark.parametrize("fold", ["1", "0"])
def test_blurred_levels_every_width_class(pkg, oracle, opts, fold):
    """cv::GaussianBlur 7x7 (ORBextractor.cc:1085-1086) on all 8 levels for level-0 widths 400..407: every w mod 4 class of the border
    dwords (BORDER_REFLECT_101 assembled in registers), through the folded kernel (k_blur_fold, DCS_BLUR_FOLD=1, the default) and
    through the round-2 pair k_blur + k_blur_edge_cols (=0)."""
    opts("DCS_BLUR_FOLD", int(fold))
    rng = np.random.default_rng(11)
    for w in range(400, 408):
        h = 300 + (w & 3)
        imgs = [rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(3)]
        e = pkg.ORBextractor(500, 1.2, 8, 20, 7, max_images=3)
        e.extract_batch(imgs)
        for i in (0, 2):
            o = oracle.OrbOracle(500, 1.2, 8, 20, 7)
            o.extract(imgs[i])
            for l in range(8):
                assert np.array_equal(e.level_image(i, l, blurred=True), oracle.gauss7_u8(o.level_image(l))), (w, i, l)
        e.close()


def test_host_batch_pipeline_equals_one_shot(pkg, oracle, synth, opts):
    """dcs_orb_extract_batch on a large host batch runs as a pipeline of image chunks (upload || kernels || download + scatter,
    csrc/orb_extract.cpp); DCS_ORB_HOST_CHUNK=0 selects the one-shot path. Same bytes either way, and the oracle's for sampled images;
    a ragged last chunk is covered."""
    W, H = 320, 240
    base = [synth.frame_pair(640, 480, 0, f) for f in range(5)]
    imgs = []
    for i in range(75):                                        # 75 images: chunks of 16 -> 4 full + 11
        a = base[i % 5][i % 2]
        y0, x0 = (7 * i) % 200, (13 * i) % 300
        imgs.append(np.ascontiguousarray(a[y0:y0 + H, x0:x0 + W]))
    outs = []
    for chunk in ("16", "0"):
        opts("DCS_ORB_HOST_CHUNK", int(chunk))
        ext = pkg.ORBextractor(300, 1.2, 8, 20, 7, max_images=75)
        outs.append(ext.extract_batch(imgs))
        ext.close()
    (kp_a, d_a), (kp_b, d_b) = outs
    for i in range(75):
        assert kp_a[i].tobytes() == kp_b[i].tobytes() and np.array_equal(d_a[i], d_b[i]), i
    o = oracle.OrbOracle(300, 1.2, 8, 20, 7)
    for i in (0, 15, 16, 63, 64, 74):
        ek, ed = o.extract(imgs[i])
        assert kp_a[i].tobytes() == ek.tobytes() and np.array_equal(d_a[i], ed), i


@pytest.mark.parametrize("W", [320, 322])
def test_page_locked_frames_are_read_in_place(pkg, oracle, synth, opts, W):
    """Frames in page-locked memory (dcs_host_alloc) at equal spacing and a 4-byte aligned stride go up without the staging copy
    (csrc/orb_extract.cpp: `direct`) -- one dual frame per call, a one-shot batch and the chunked pipeline; same bytes as pageable
    arrays of the same pixels (DCS_ORB_HOST_DIRECT=0: the packing path on the very same pointers) and the oracle's. W = 322: the
    stride (324) is wider than the rows."""
    H = 240
    base = [synth.frame_pair(640, 480, 0, f) for f in range(5)]
    hf = pkg.abi.HostFrames(40, H, W)
    assert hf.stride % 4 == 0 and hf.stride >= W
    imgs = []
    for i in range(40):
        a = base[i % 5][i % 2]
        y0, x0 = (11 * i) % 200, (17 * i) % 300
        hf.frames[i][:] = a[y0:y0 + H, x0:x0 + W]
        imgs.append(hf.frames[i].copy())
    ext = pkg.ORBextractor(300, 1.2, 8, 20, 7, max_images=40)
    ref = ext.extract_batch(imgs)                                             # pageable copies: the packing path
    for chunk, n in (("0", 2), ("0", 40), ("8", 40)):
        opts("DCS_ORB_HOST_CHUNK", int(chunk))
        for direct in ("1", "0"):
            opts("DCS_ORB_HOST_DIRECT", int(direct))
            kp, d = ext.extract_batch(hf.frames[:n], stride=hf.stride)
            for i in range(n):
                assert kp[i].tobytes() == ref[0][i].tobytes() and np.array_equal(d[i], ref[1][i]), (chunk, n, direct, i)
    # frames that are NOT equally spaced (a reordered ring) fall back to the packing path
    order = [3, 1, 2]
    kp, d = ext.extract_batch([hf.frames[i] for i in order], stride=hf.stride)
    for j, i in enumerate(order):
        assert kp[j].tobytes() == ref[0][i].tobytes() and np.array_equal(d[j], ref[1][i])
    ext.close()
    o = oracle.OrbOracle(300, 1.2, 8, 20, 7)
    for i in (0, 1, 39):
        ek, ed = o.extract(imgs[i])
        assert ref[0][i].tobytes() == ek.tobytes() and np.array_equal(ref[1][i], ed), i
    hf.close()


def test_separately_pinned_frames_are_not_read_as_one_range(pkg, synth, opts):
    """A left and a right frame that were pinned SEPARATELY (two dcs_host_alloc blocks, two hipHostRegister'd cv::Mats) are 'equally spaced'
    by construction, but the memory between them belongs to neither: the in-place path (one DMA over the whole range) is only taken when
    the range lies inside ONE page-locked allocation, anything else is packed. Same features either way."""
    H, W = 240, 320
    a, b = synth.frame_pair(640, 480, 2, 0)
    one = pkg.abi.HostFrames(2, H, W)
    left, right = pkg.abi.HostFrames(1, H, W), pkg.abi.HostFrames(1, H, W)
    pad = pkg.abi.HostFrames(1, 7 * H, W)                                # (keeps the allocator from handing out two adjacent blocks)
    for dst, src in ((one.frames[0], a), (one.frames[1], b), (left.frames[0], a), (right.frames[0], b)):
        dst[:] = src[100:100 + H, 150:150 + W]
    opts("DCS_ORB_SMALL_GRAPH", int("0"))
    ext = pkg.ORBextractor(400, 1.2, 8, 20, 7, max_images=2)
    ref = ext.extract_batch([one.frames[0].copy(), one.frames[1].copy()])
    assert ext.host_path()[0] == 0                                        # pageable copies: packed
    kp, d = ext.extract_batch(one.frames, stride=one.stride)
    assert ext.host_path()[0] == 1                                        # one block: read in place
    for c in range(2):
        assert kp[c].tobytes() == ref[0][c].tobytes() and np.array_equal(d[c], ref[1][c])
    for pair in ([left.frames[0], right.frames[0]], [right.frames[0], left.frames[0]]):
        order = (0, 1) if pair[0] is left.frames[0] else (1, 0)
        kp, d = ext.extract_batch(pair, stride=left.stride)
        assert ext.host_path()[0] == 0, "two allocations were read as one range"
        for j, c in enumerate(order):
            assert kp[j].tobytes() == ref[0][c].tobytes() and np.array_equal(d[j], ref[1][c])
    ext.close()
    for hf in (one, left, right, pad):
        hf.close()


def test_small_call_graph_is_dropped_when_the_handle_is_reconfigured(pkg, synth, opts):
    """The executable graph of a 1-2 image call holds the addresses of the handle's internal buffers; a call of ANOTHER shape through the
    _device entry point rebuilds them. The next host call of the first shape must not replay the old graph: it is captured
    anew and every call returns the features of its own frames."""
    import torch
    opts("DCS_ORB_SMALL_GRAPH", int("1"))
    frames = [synth.frame_pair(640, 480, 3, f) for f in range(4)]
    small = [[np.ascontiguousarray(im[60:300, 80:400]) for im in fp] for fp in frames]
    opts("DCS_ORB_SMALL_GRAPH", int("0"))
    r = pkg.ORBextractor(500, 1.2, 8, 20, 7, max_images=2)
    ref = [r.extract_batch(s_) for s_ in small]
    r.close()
    opts("DCS_ORB_SMALL_GRAPH", int("1"))
    e = pkg.ORBextractor(500, 1.2, 8, 20, 7, max_images=2)
    for i in range(4):
        kp, d = e.extract_batch(small[i])
    assert e.host_path()[1] == 1                                          # replayed by now
    cap = e.default_cap()
    big = torch.from_numpy(np.stack(frames[0])).cuda()                    # 640 x 480: every internal buffer grows
    d_kp = torch.zeros((2, cap, 7), dtype=torch.float32, device="cuda"); d_desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(2, dtype=torch.int32, device="cuda")
    e.extract_batch_device(big, d_kp, d_desc, d_n, cap)
    torch.cuda.synchronize()
    for rep in range(2):
        for i in range(4):
            kp, d = e.extract_batch(small[i])
            if rep == 0 and i == 0:
                assert e.host_path()[1] == 0, "a graph captured before the reconfiguration was replayed"
            for c in range(2):
                assert kp[c].tobytes() == ref[i][0][c].tobytes() and np.array_equal(d[c], ref[i][1][c]), (rep, i, c)
    assert e.host_path()[1] == 1
    e.close()


def test_one_frame_calls_replayed_as_a_graph(pkg, oracle, synth, opts):
    """A handle that is called with one or two host images again and again (Frame::ExtractORB, src/Frame.cc:141-149) replays the call's
    launches as one executable graph from the third call of a shape on (DCS_ORB_SMALL_GRAPH=1): every call still returns the features
    of ITS images -- different frames every call, compared with a fresh handle without the graph and with the oracle."""
    frames = [synth.frame_pair(640, 480, 1, f) for f in range(7)]
    opts("DCS_ORB_SMALL_GRAPH", int("0"))
    ref_ext = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=2)
    ref = [ref_ext.extract_batch(list(fp)) for fp in frames]
    ref_ext.close()
    opts("DCS_ORB_SMALL_GRAPH", int("1"))
    e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=2)
    for rep in range(2):
        for i, fp in enumerate(frames):
            kps, descs = e.extract_batch(list(fp))
            for c in range(2):
                assert kps[c].tobytes() == ref[i][0][c].tobytes() and np.array_equal(descs[c], ref[i][1][c]), (rep, i, c)
    one = e.extract_batch([frames[3][1]])                         # another shape of call on the same handle: single image
    assert one[0][0].tobytes() == ref[3][0][1].tobytes()
    kps, descs = e.extract_batch(list(frames[5]))                  # and back
    assert kps[0].tobytes() == ref[5][0][0].tobytes() and np.array_equal(descs[1], ref[5][1][1])
    e.close()
    o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    ek, ed = o.extract(frames[2][0])
    assert ref[2][0][0].tobytes() == ek.tobytes() and np.array_equal(ref[2][1][0], ed)


def test_fast_hw_probe_and_fallback_are_bit_exact(pkg, oracle, synth, opts):
    """the start-up probe (k_fast_hw_probe) accepts this device's ds_read_u8_d16_hi / v_cmpx behaviour; a handle created while the probe is made
    to fail (DCS_FAST_HW_PROBE=fail) runs k_fast_cells' plain forms -- byte loads, ballot append -- and returns the same bytes"""
    imgs = list(synth.frame_pair(640, 480, 6, 1))
    e = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=2)
    hw = e.fast_hw()
    kps, descs = e.extract_batch(imgs)
    e.close()
    opts("DCS_FAST_HW_PROBE_FAIL", 1)
    e2 = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_images=2)
    assert e2.fast_hw() == 0
    kps2, descs2 = e2.extract_batch(imgs)
    e2.close()
    assert hw == 1, "the probe rejected this device: the fast forms are off everywhere (results are still exact)"
    for i in range(2):
        _same(kps[i], descs[i], kps2[i], descs2[i])
        okp, odesc = oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(imgs[i])
        _same(kps2[i], descs2[i], okp, odesc)
