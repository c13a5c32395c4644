from .bounding_box import BoxList
from .image_list import ImageList, to_image_list
from .boxlist_ops import boxlist_nms, boxlist_iou, cat_boxlist, remove_small_boxes

__all__ = ["BoxList", "ImageList", "to_image_list", "boxlist_nms", "boxlist_iou", "cat_boxlist",
           "remove_small_boxes"]
