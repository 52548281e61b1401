#!/usr/bin/env python
"""Row (f1): `.npy` feature files -> test-mode tiles resident in HBM (anomalyclip_amd.feature_stream.FeatureStream: one
vectorised gather into pinned memory + an asynchronous copy on a side stream, double-buffered), timed (a) alone, (b) feeding
the UCF head's test forward, against (c) the reference's per-frame Python loop (feature_dataset.py:359-367: 512*S tensor
indexings + torch.cat per video) on the host.  Files live in the page cache (written just before), so the numbers are the
loader's, not the disk's.  One JSON line.   python tools/bench_feature_stream.py [--videos 32] [--ref-videos 4]"""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def reference_loop(arr, N=32, L=16):
    """feature_dataset.py:243-259,359-367 restated as the reference runs it (test mode, stride 1, one crop)."""
    T = arr.shape[0]
    feats = torch.from_numpy(arr)
    starts = np.arange(np.ceil(T / (N * L)) * N * L / L) * L
    out = []
    for s in starts:
        for i in range(L):
            out.append(feats[(int(s) + i) % T].unsqueeze(0))
    return torch.cat(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=32)
    ap.add_argument("--ref-videos", type=int, default=4)
    args = ap.parse_args()
    import bench as B
    from anomalyclip_amd.feature_stream import FeatureStream
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    lengths = [int(v) for v in rng.integers(1500, 9000, size=args.videos)]       # UCF-Crime test videos: a few thousand frames
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for i, T in enumerate(lengths):
            p = os.path.join(td, f"v{i}.npy")
            np.save(p, (rng.standard_normal((T, 512)) * 0.3).astype(np.float32))
            paths.append(p)
        tile_rows = sum(-(-T // 512) * 512 for T in lengths)
        # (a) loader alone
        fs = FeatureStream(paths, device=dev)
        for _ in fs:                                       # warm pass: pins the two host buffers at their final size
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for feats, T, S, path in fs:
            pass
        torch.cuda.synchronize()
        dt_load = time.perf_counter() - t0
        # (b) loader feeding the head
        net, sd, eot, hc = B.build_net("f32", dev)
        net.load_from_features = True                                               # the head alone: features in, scores out
        net.cache_text_features = False                                             # (b): the reference's per-video text tower
        nc = torch.zeros(512, device=dev)
        with torch.no_grad():
            for feats, T, S, path in fs:
                net(feats, None, nc, S, True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for feats, T, S, path in fs:
                net(feats, None, nc, S, True)
            torch.cuda.synchronize()
            dt_score = time.perf_counter() - t0
            # (b') the same with the text features kept across test steps (AnomalyCLIP(cache_text_features=True): the prompt
            # parameters are frozen in evaluation, so the per-video text tower of the reference recomputes the same tensor)
            net.cache_text_features = True
            for feats, T, S, path in fs:
                net(feats, None, nc, S, True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for feats, T, S, path in fs:
                net(feats, None, nc, S, True)
            torch.cuda.synchronize()
            dt_cached = time.perf_counter() - t0
            # (b'') several videos per forward (FeatureStream.batched -> AnomalyCLIP.forward_test_many): one launch sequence
            # and one text-tower evaluation per group of videos; text features NOT cached across groups, then cached
            def run_batched(videos, tiles):
                for feats, meta in fs.batched(videos=videos, max_tiles=tiles):
                    net.forward_test_many(feats, [m[2] for m in meta], [m[1] for m in meta], nc)
            batched = {}
            for cached in (False, True):
                net.cache_text_features = cached
                run_batched(8, 96)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run_batched(8, 96)
                torch.cuda.synchronize()
                batched[cached] = time.perf_counter() - t0
            net.cache_text_features = False
            # loader alone, batched (4 reader threads per group), and the head alone on resident features (no loader)
            for _ in fs.batched(videos=8, max_tiles=96):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in fs.batched(videos=8, max_tiles=96):
                pass
            torch.cuda.synchronize()
            dt_load_b = time.perf_counter() - t0
            groups = [(f.clone(), m) for f, m in fs.batched(videos=8, max_tiles=96)]
            for f, m in groups:
                net.forward_test_many(f, [q[2] for q in m], [q[1] for q in m], nc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f, m in groups:
                net.forward_test_many(f, [q[2] for q in m], [q[1] for q in m], nc)
            torch.cuda.synchronize()
            dt_head_b = time.perf_counter() - t0
        # (c) the reference's loop on the host
        t0 = time.perf_counter()
        ref_rows = 0
        for p in paths[: args.ref_videos]:
            ref_rows += reference_loop(np.load(p)).shape[0]
        dt_ref = time.perf_counter() - t0
    out = {"row": "f1 feature files -> HBM tiles", "videos": args.videos, "tile_rows": tile_rows, "tile_bytes": tile_rows * 2048,
           "loader": {"ms": round(dt_load * 1e3, 2), "features_per_s": round(tile_rows / dt_load, 1),
                      "GBps_into_hbm": round(tile_rows * 2048 / dt_load / 1e9, 2)},
           "loader_plus_head_test_forward": {"ms": round(dt_score * 1e3, 2), "features_per_s": round(tile_rows / dt_score, 1)},
           "loader_plus_head_cached_text_features": {"ms": round(dt_cached * 1e3, 2), "features_per_s": round(tile_rows / dt_cached, 1)},
           "batched_8_videos_per_forward": {"ms": round(batched[False] * 1e3, 2), "features_per_s": round(tile_rows / batched[False], 1),
                                            "note": "FeatureStream.batched(8 videos, <= 96 tiles) -> forward_test_many; text tower once per group"},
           "batched_loader_alone": {"ms": round(dt_load_b * 1e3, 2), "features_per_s": round(tile_rows / dt_load_b, 1),
                                    "GBps_into_hbm": round(tile_rows * 2048 / dt_load_b / 1e9, 2)},
           "batched_head_alone_resident_features": {"ms": round(dt_head_b * 1e3, 2), "features_per_s": round(tile_rows / dt_head_b, 1),
                                                    "note": "forward_test_many on device-resident groups, text tower once per group (uncached)"},
           "batched_8_videos_cached_text_features": {"ms": round(batched[True] * 1e3, 2), "features_per_s": round(tile_rows / batched[True], 1)},
           "cpu_baseline": {"features_per_s": round(ref_rows / dt_ref, 1), "cores": 1, "kind": "reference loop (restated)",
                            "sample": f"{args.ref_videos} videos, {ref_rows} rows"}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
