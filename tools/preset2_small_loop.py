import os, sys
root = "/root/repo"
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import synth
from pixo_amd import jpeg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
px = synth.noise(n, n, 42)
for _ in range(200): jpeg.encode_jpeg(px, n, n, 2, 80, 2, True)
