"""The 80 COCO object categories in the index order YOLOv7 emits (the reference keeps the same list in
vlfm/vlm/coco_classes.py and routes a target to YOLOv7 iff its name is in it, base_objectnav_policy.py:221-223), and the
extra MP3D category names of vlfm/vlm/classes.txt.

COCO's index order is its supercategory order, so the table is kept by supercategory and flattened."""

_COCO_BY_SUPERCATEGORY = {
    "person": "person",
    "vehicle": "bicycle car motorcycle airplane bus train truck boat",
    "outdoor": "traffic light|fire hydrant|stop sign|parking meter|bench",
    "animal": "bird cat dog horse sheep cow elephant bear zebra giraffe",
    "accessory": "backpack umbrella handbag tie suitcase",
    "sports": "frisbee|skis|snowboard|sports ball|kite|baseball bat|baseball glove|skateboard|surfboard|tennis racket",
    "kitchen": "bottle|wine glass|cup|fork|knife|spoon|bowl",
    "food": "banana|apple|sandwich|orange|broccoli|carrot|hot dog|pizza|donut|cake",
    "furniture": "chair|couch|potted plant|bed|dining table|toilet",
    "electronic": "tv|laptop|mouse|remote|keyboard|cell phone",
    "appliance": "microwave oven toaster sink refrigerator",
    "indoor": "book|clock|vase|scissors|teddy bear|hair drier|toothbrush",
}
COCO_CLASSES = [name for group in _COCO_BY_SUPERCATEGORY.values() for name in group.split("|" if "|" in group else " ")]
assert len(COCO_CLASSES) == 80 and COCO_CLASSES[56] == "chair" and COCO_CLASSES[79] == "toothbrush"

MP3D_EXTRA_CLASSES = ("framed photograph|cabinet|pillow|nightstand|sink|stool|towel|shower|bathtub|counter|fireplace|"
                      "gym equipment|seating|clothes|cupboard|table").split("|")
