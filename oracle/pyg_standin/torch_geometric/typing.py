from typing import Optional
from torch import Tensor

OptTensor = Optional[Tensor]
