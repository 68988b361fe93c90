"""End-to-end training throughput from image FILES: DeNet-34 skip 512x512, batch 32, 640x480 JPEGs on disk, denet crop +
photometric jitter planned on the host and rendered on the GPU (denet_amd/dataset/device_render.py), the next batch
prepared on a side stream while the current one trains. Compare with bench.py (batch resident in HBM)."""
import os, sys, time, random, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy, torch
import dataset_scenarios as S
from denet_amd.dataset.device_render import DeviceImageLoader
from denet_amd.model import zoo


def main():
    THREADS = int(os.environ.get("THREADS", 8))
    NIMG = int(os.environ.get("NIMG", 320))
    B = 32
    with tempfile.TemporaryDirectory() as root:
        images = []
        for i in range(16):
            f = os.path.join(root, "im%d.jpg" % i)
            w, h = ((640, 480), (480, 640), (640, 427), (500, 375))[i % 4]
            S.synth_image(i, w, h).save(f, format="JPEG", quality=90)
            images.append({"fname": f, "bboxs": S.synth_boxes(i, w, h, 7, 80), "id": i})
        images = [images[i % 16] for i in range(NIMG)]
        random.seed(1); numpy.random.seed(1)
        model = zoo.denet34(B, "skip", 512, class_num=80, seed=1)
        model.build_train_func("nesterov")
        loader = DeviceImageLoader(THREADS, True, {"crop": 512, "crop_mode": "denet", "augment_photo": True, "check_center": True},
                                   decode=os.environ.get("DECODE", "process"))
        model.train_epoch_device(loader, images[:3 * B], 0, 0.01, [0.9], 1e-4)       # warm-up (autotune, pools)
        torch.cuda.synchronize()
        t0 = time.time()
        cost = model.train_epoch_device(loader, images, 0, 0.01, [0.9], 1e-4)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("real-data training: %d images in %.2f s = %.1f img/s (%.1f ms per batch of %d), %d decode threads, cost/batch %.3f" % (
            NIMG, dt, NIMG / dt, 1e3 * dt / (NIMG / B), B, THREADS, cost / (NIMG / B)), os.environ.get("DECODE", "process"))
        loader.close()


if __name__ == "__main__":      # the decode workers are spawned: they re-import this module
    main()
