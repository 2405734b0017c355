"""cProfile of a few SI iterations at the bench shapes: where the HOST time goes."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
model, X, Y = build_model(2000, 5, 100, 0)
for _ in range(3):
    model.imp.sample(burnin=10); model._m_step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    model.imp.sample(burnin=10); model._m_step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumtime').print_stats(60)
