"""MI355X-native UNet segmentation train-step path (hand-written HIP kernels behind the reference's
nn.Module / loss-function surface).  See DESIGN.md and include/unet_hip.h."""
from .unet import UNet, UNet_S, UNet_T, UNetDepth, DoubleConv, Down, Up, OutConv  # noqa: F401
from .utils.dice_score import dice_coeff, multiclass_dice_coeff, dice_loss  # noqa: F401
from .utils.boundary_loss import boundary_loss  # noqa: F401
from .utils.connected_component_loss import connected_component_loss  # noqa: F401
from .train import FusedRMSprop, seg_loss, train_step, TrainStepper, GraphedTrainStepper  # noqa: F401
from .evaluate import evaluate  # noqa: F401
from .predict import predict_img, mask_to_image, preprocess_image  # noqa: F401
from .checkpoint import save_checkpoint, load_checkpoint  # noqa: F401
from .synthetic import ellipse_batch  # noqa: F401
from .utils.data_loading import BasicDataset, CarvanaDataset, load_image  # noqa: F401
from .utils.post_process import postprocess_mask, remove_internal_regions  # noqa: F401
from .inference import GraphedForward  # noqa: F401
