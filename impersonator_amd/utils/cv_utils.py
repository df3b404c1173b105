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


def euler2matrix(rt):
    """utils/cv_utils.py:333-353: R = Rz @ Ry @ Rx from euler angles (3,)."""
    cx, sx = np.cos(rt[0]), np.sin(rt[0])
    cy, sy = np.cos(rt[1]), np.sin(rt[1])
    cz, sz = np.cos(rt[2]), np.sin(rt[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=np.float32)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float32)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=np.float32)
    return np.dot(Rz, np.dot(Ry, Rx))
