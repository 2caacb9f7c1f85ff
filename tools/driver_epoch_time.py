"""Wall time of one training epoch (469 steps of batch 128 on the 60 000 synthetic images) THROUGH THE DROP-IN DRIVER: (T(5 epochs) - T(1 epoch)) / 4."""
import os, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = "/tmp/vd_drv"
shutil.rmtree(R, ignore_errors=True)
os.makedirs(R)


def run(ep):
    t0 = time.perf_counter()
    subprocess.run([sys.executable, os.path.join(ROOT, "VillanDiffusion.py"), "--project", "default", "--mode", "train", "--dataset", "SYNTHETIC-CIFAR10", "--batch", "128",
                    "--epoch", str(ep), "--poison_rate", "0.1", "--trigger", "BOX_14", "--target", "HAT", "--ckpt", "DDPM-32-DEFAULT", "--fclip", "o", "-o",
                    "--sched", "DDIM-SCHED", "--infer_steps", "10", "--save_image_epochs", "100", "--save_model_epochs", "100", "--result", f"{R}/e{ep}"],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    return time.perf_counter() - t0


t1, t5 = run(1), run(5)
e = (t5 - t1) / 4
print(f"1-epoch run {t1:.1f} s, 5-epoch run {t5:.1f} s -> {e:.2f} s per epoch of 469 steps = {e / 469 * 1e3:.2f} ms/step = {60000 / e:.0f} img/s through the driver")
