import sys; sys.path.insert(0, '/root/repo' if __import__('os').path.exists('/root/repo/tools') else '.')
sys.path.insert(0, '.')
import tools.gemm8_bench as g
from emip_amd import _lib
_lib.load()
cf = [1, 2, 3, 8, 9]
for M, N, K in [(7744, 1280, 320), (7744, 320, 1280), (7744, 320, 320), (30976, 512, 128), (30976, 128, 512), (123904, 256, 64), (123904, 64, 256), (30976, 1024, 256), (30976, 128, 128), (1936, 640, 320), (15488, 1280, 320)]:
    g.dense(M, N, K, cf)
