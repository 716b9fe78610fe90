"""Device-side input transform rate: 32 clips of 8 x 360 x 640 uint8 frames -> [32,3,8,224,224] bf16 (train transform)."""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, youku_mplug_amd
from youku_mplug_amd.video_input import VideoInputTransform
dev = torch.device("cuda:0")
clips = [torch.randint(0, 256, (8, 360, 640, 3), dtype=torch.uint8, device=dev) for _ in range(32)]
tf = VideoInputTransform(224, train=True)
random.seed(0)
for _ in range(3):
    out = tf.batch(clips)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    out = tf.batch(clips)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
inb = 32 * 8 * 360 * 640 * 3
outb = out.numel() * 2
print(f"batch of 32 clips: {dt*1e3:.3f} ms  -> {32/dt:.0f} clips/s; bytes in {inb/1e6:.0f} MB (crop-dependent) out {outb/1e6:.0f} MB, {(inb+outb)/dt/1e9:.0f} GB/s upper-bound traffic")
