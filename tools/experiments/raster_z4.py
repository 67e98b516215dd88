import itertools, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import refrun
from tools.experiments.raster_z3 import *   # reuses the loop's last state is not what we want; re-run selected triangle below
