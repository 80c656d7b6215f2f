"""End-to-end forward under rocprofv3 --kernel-trace --stats: warm-up first (MIOpen first-use work), then 20 batches."""
import sys, torch
sys.path.insert(0, '.')
import snn_automotive_object_detection_amd as S
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = S.create_model('cityscapes', 9, True, True, 0, False, False, 8, 12).to(dev).eval()
imgs = [torch.rand((3, 1024, 2048), device=dev) for _ in range(2)]
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
        m(imgs)
torch.cuda.synchronize()
