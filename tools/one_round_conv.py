"""conv+LIF launches of exactly ONE round of work-groups (2 per CU start together and stay in phase) and of 4 / 24 rounds:
TCC_READ_sum per work-group tells whether the per-CU L1 serves the second work-group's weight chunks while the two are in phase
(rocprofv3 --kernel-trace --pmc TCC_READ_sum -- python3 tools/one_round_conv.py)."""
import sys, torch
sys.path.insert(0, '.')
import snn_automotive_object_detection_amd as S
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = S.RPNHeadSNN(256, 3, 8).to(dev)
for h, w in ((64, 64), (128, 128), (256, 384)):          # 2 x h x w positions / 32 per tile x 2 column blocks = 512 / 2048 / 12288 work-groups
    f = [torch.randn(2, 256, h, w, device=dev)]
    for _ in range(3):
        m(f)
    torch.cuda.synchronize()
    print(h, w, "work-groups", 2 * h * w // 32 * 2)
