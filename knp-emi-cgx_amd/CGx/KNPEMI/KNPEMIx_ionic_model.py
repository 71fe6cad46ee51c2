"""Same import path as the reference's src/CGx/KNPEMI/KNPEMIx_ionic_model.py."""
from cgx_hip.ionic_models import (ATPPump, GlialCotransporters, HodgkinHuxley, IonicModel,  # noqa: F401
                                  KirNaKPumpModel, NeuronalCotransporters, PassiveModel)
