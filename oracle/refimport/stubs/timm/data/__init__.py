"""Stand-in for the two constants of the un-vendored `timm` package that WaveCAM-TMM2023/net/wavecam.py:7 imports (the module is
pulled in by net/resnet50_cam.py:151; the CAM path of config 5 does not use them).  Published values of timm.data.constants."""
IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
