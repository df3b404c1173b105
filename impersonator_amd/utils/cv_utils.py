"""Image IO at the host boundary (utils/cv_utils.py of the reference, which uses OpenCV; PIL here)."""
import numpy as np


def read_cv2_img(path):
    """utils/cv_utils.py:6-17: RGB uint8 (H,W,3)."""
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def transform_img(image, image_size, transpose=False):
    """utils/cv_utils.py:36-44: resize to (image_size, image_size), scale to [0,1], optional CHW."""
    from PIL import Image
    if image.shape[0] != image_size or image.shape[1] != image_size:
        image = np.asarray(Image.fromarray(image).resize((image_size, image_size), Image.BILINEAR))
    image = image.astype(np.float32) / 255.
    if transpose:
        image = image.transpose((2, 0, 1))
    return image


def save_cv2_img(img, path, image_size=None, normalize=False):
    """utils/cv_utils.py:20-34: truncating uint8 conversion of [-1,1] images (hazard H11)."""
    from PIL import Image
    if normalize:
        img = (img + 1) / 2.0 * 255
        img = img.astype(np.uint8)
    if image_size is not None:
        img = np.asarray(Image.fromarray(img).resize((image_size, image_size), Image.BILINEAR))
    Image.fromarray(img).save(path)
    return img
