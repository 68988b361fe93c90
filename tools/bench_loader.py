"""Throughput of the host data pipeline (denet_amd/dataset) for the headline configuration: MSCOCO-sized images
(640x480), crop_mode=denet to 512x512, photometric jitter, per worker process. Compares with the training rate the GPU
sustains (bench.py) to size the loader pool: cores needed per GPU = train img/s / loader img/s per core."""
import os, sys, time, tempfile, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy
from denet_amd.dataset import image_loader
import dataset_scenarios as S


def main():
    N = int(os.environ.get("N", 48))
    with tempfile.TemporaryDirectory() as root:
        images = []
        for i in range(8):
            f = os.path.join(root, "im%d.jpg" % i)
            S.synth_image(i, 640, 480).save(f, format="JPEG", quality=90)
            images.append({"fname": f, "bboxs": S.synth_boxes(i, 640, 480, 7, 80), "id": i})
        for threads in (1, 4):
            for mode, extra in (("denet", {}), ("denet", {"augment_photo": True}), ("default", {"scale": 512})):
                fp = {"crop": 512, "crop_mode": mode}
                fp.update(extra)
                loader = image_loader.ImageLoader(threads, True, fp)
                random.seed(1)
                loader.load(images[:threads])       # warm the pool
                t0 = time.time()
                data = loader.load([images[i % 8] for i in range(N)])
                dt = time.time() - t0
                assert len(data) == N and data[0][1].shape == (3, 512, 512)
                print("workers %d crop_mode %-8s %-22s: %6.1f img/s (%5.1f ms per image per worker)" % (
                    threads, mode, str(extra), N / dt, 1e3 * dt * threads / N), flush=True)
                loader.close()

        # device rendering: decode on threads, everything else on the GPU (needs a GPU)
        try:
            import torch
            have_gpu = torch.cuda.is_available()
        except Exception:
            have_gpu = False
        if have_gpu:
            from denet_amd.dataset.device_render import DeviceImageLoader
            for threads in (1, 4, 8):
                for extra in ({}, {"augment_photo": True}):
                    fp = {"crop": 512, "crop_mode": "denet"}
                    fp.update(extra)
                    loader = DeviceImageLoader(threads, True, fp)
                    random.seed(1)
                    batch = [images[i % 8] for i in range(32)]
                    loader.load_batch(batch); torch.cuda.synchronize()
                    t0 = time.time()
                    for _ in range(4):
                        x, metas = loader.load_batch(batch)
                    torch.cuda.synchronize()
                    dt = time.time() - t0
                    print("device render, decode threads %d %-22s: %6.1f img/s (batch of 32 in %.1f ms)" % (
                        threads, str(extra), 128 / dt, 1e3 * dt / 4), flush=True)


if __name__ == "__main__":      # loader workers are spawned: they re-import this module
    main()
