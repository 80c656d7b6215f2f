"""Run the stock backbone alone (for rocprofv3 --kernel-trace --stats): which MIOpen / rocBLAS kernels carry its 10.7 ms."""
import sys, torch
sys.path.insert(0, '.')
import snn_automotive_object_detection_amd as S
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = S.create_model('cityscapes', 9, True, True, 0, False, False, 8, 12).to(dev).eval()
imgs = [torch.rand((3, 1024, 2048), device=dev) for _ in range(2)]
with torch.no_grad():
    il, _ = m.transform(imgs)
    for _ in range(10):
        m.backbone(il.tensors)
torch.cuda.synchronize()
