import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf-keras-speech-commands_amd"))
import torch
from classifier.params import pr
from kws_amd.featurizer import Featurizer
f = Featurizer(pr)
wav = (0.1 * torch.randn((4096, 16000), device="cuda")).contiguous()
out = torch.empty((4096, 30, 20), device="cuda")
for _ in range(3): f(wav, out=out)
torch.cuda.synchronize()
