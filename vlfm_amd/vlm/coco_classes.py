"""COCO-80 class names in YOLOv7 index order (data; same list as /root/reference/vlfm/vlm/coco_classes.py) and the extra
MP3D class names of /root/reference/vlfm/vlm/classes.txt."""

COCO_CLASSES = [
    "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light",
    "fire hydrant", "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow", "elephant",
    "bear", "zebra", "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee", "skis", "snowboard",
    "sports ball", "kite", "baseball bat", "baseball glove", "skateboard", "surfboard", "tennis racket", "bottle",
    "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple", "sandwich", "orange", "broccoli",
    "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch", "potted plant", "bed", "dining table", "toilet",
    "tv", "laptop", "mouse", "remote", "keyboard", "cell phone", "microwave", "oven", "toaster", "sink",
    "refrigerator", "book", "clock", "vase", "scissors", "teddy bear", "hair drier", "toothbrush",
]

MP3D_EXTRA_CLASSES = [
    "framed photograph", "cabinet", "pillow", "nightstand", "sink", "stool", "towel", "shower", "bathtub", "counter",
    "fireplace", "gym equipment", "seating", "clothes", "cupboard", "table",
]
