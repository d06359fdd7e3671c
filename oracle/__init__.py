"""CPU oracle: a plain PyTorch-fp32 / numpy restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `real_esrgan-pytorch_amd/` may import this package;
only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and
only as the checker / reported baseline, never as the thing shipped or measured.

Parity status: pinned against outputs of the reference itself (imported in the build
container with torchvision/cv2 stand-ins, see tests/golden/ref_shim.py) through the golden
vectors committed under tests/golden/*.npz (generator: tests/golden/gen_golden.py).
Three third-party formulas are "pinned to formula, unpinned vs the real library":
rgb_to_grayscale weights, cv2.getGaussianKernel, torchvision.to_tensor.
"""
