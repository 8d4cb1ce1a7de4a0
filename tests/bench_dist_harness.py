"""One rank of bench.py's distributed path on CPU: the oracle backend stands in for the HIP libraries
(injected here, by the test — bench.py itself never loads it), gloo for RCCL.  Launched by
tests/test_bench_distributed.py through torch.distributed.run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import harness as H  # noqa: E402
import bench  # noqa: E402

if __name__ == "__main__":
    bench.main(sys.argv[1:], backend=H.oracle_backend(), tensor_device="cpu")
