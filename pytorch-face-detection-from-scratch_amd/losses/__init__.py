from .YoloLoss import yolo_loss, yolo_loss_batch  # noqa: F401
