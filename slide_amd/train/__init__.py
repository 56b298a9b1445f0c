"""Training step of the latent DDPMs on the HIP row-major path.

functions.py  torch.autograd.Functions over the module path's forward kernels and the backward kernels of csrc/train_ops.hip
              (include/slide_train.h): 1x1 convolution / linear, MyGroupNorm (+ ReLUs), grouping, relu([q | k]), softmax-weighted sum
denoiser.py   PointNet2CloudCondition (pointnet2/models/pointnet2_with_pcld_condition.py:286-489) built from them, with the
              reference's parameter names
losses.py     util.training_loss (pointnet2/util.py:262-300) and LatentDiffusion.train_loss
              (pointnet2/diffusion_utils/diffusion.py:319-341)
dp.py         data-parallel gradient averaging: bucketed all-reduce over torch.distributed (RCCL over xGMI; gloo in the CPU tests),
              the counterpart of pointnet2/distributed.py:99-151
There is no CPU fallback: every Function launches kernels of libslide_hip.so."""
