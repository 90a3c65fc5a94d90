"""Phase stamps of the token engine (worker 0, one layer of one position): MIS_TE_STAMPS=<position> python tools/token_engine_stamps.py [xcds]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mlx_audio_swift_amd as mas
os.environ.setdefault("MIS_TE_STAMPS", "80")
cfg = mas.SopranoConfiguration(stop_token_id=-1)
lm = mas.LlamaTTSModel.synthetic(cfg.lm_configuration(), seed=4321)
prompt = np.random.default_rng(1235).integers(4, 8000, 24).astype(np.int32)
for rep in range(2):
    r = lm.debug_token_engine(prompt, 64, xcds=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
print("ms", r["ms"], "us per position", r["ms"] * 1e3 / 88)
