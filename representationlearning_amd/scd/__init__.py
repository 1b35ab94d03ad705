"""SCD (AAAI 2023) class-activation-map path on the librssf kernels: the Mix-Transformer encoder, the CAM / attention heads of
`TSCD(..., cam_only=True)` and `multi_scale_cam` - BASELINE config 5 as it is worded (reference: /SCD-AAAI2023/network,
/SCD-AAAI2023/utils/camutils.py).  Inference only, like the reference's CAM extraction (torch.no_grad)."""
