import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf-keras-speech-commands_amd"))
import torch
from classifier.params import pr
from kws_amd.featurizer import Featurizer
print("blocks/CU, LDS bytes:", Featurizer(pr).occupancy())
