from .seal_utils import SealBBoxMapper, get_seal_mapper  # noqa: F401
from .renderer import SealTeacherMixin, make_teacher, make_student  # noqa: F401
from .trainer import GraphedSealTrainer, SealTrainer, get_trainer, sample_points  # noqa: F401
from .provider import SealDataset  # noqa: F401
