"""WaveCAM (TMM 2023) ResNet-50 CAM inference path on the librssf kernels - the conv-only relative of BASELINE config 5
(the "ResNet-38d" the config names does not exist in the reference)."""
