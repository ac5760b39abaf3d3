import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, warnings
from rpgp_amd import settings, linear_cg as lcg
from rpgp_amd.training import create_exact_gp
from rpgp_amd.models import ExactMarginalLogLikelihood
dev = torch.device("cuda:0")
N, d, J, ntest = 391386, 3, 3, 43488
g = torch.Generator().manual_seed(0)
X = torch.randn(N + ntest, d, generator=g)
y = torch.sin(X).sum(1) + 0.05 * torch.randn(N + ntest, generator=g)
y = (y - y.mean()) / y.std()
Xtr, ytr, Xte, yte = X[:N].to(dev), y[:N].to(dev), X[N:].to(dev), y[N:].to(dev)
torch.manual_seed(0); np.random.seed(0)
model, lik = create_exact_gp(Xtr, ytr, "additive_rp", J=J, noise_prior=True, kernel_type="RBF", learn_proj=False, prescale=True, space_proj=True, ski=True, ski_options={"grid_size": 1024, "num_dims": 1})
model = model.to(dev)
mll = ExactMarginalLogLikelihood(lik, model)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.1)
with settings.cg_tolerance(0.05), settings.eval_cg_tolerance(0.01), settings.max_cg_iterations(10000):
    for it in range(11):
        model.train(); lcg.stats["iterations"] = 0
        opt.zero_grad(); loss = -mll(model(Xtr), ytr); loss.backward(); opt.step()
        with torch.no_grad():
            model.eval()
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                with settings.skip_posterior_variances(True):
                    out = model(Xte)
            rmse = float(((out.mean - yte) ** 2).mean().sqrt())
        print("step %d loss %.4f train_cg_iters %d noise %.4g outputscale %.4g ls %s | pred rmse %.4f pred_iters %d warnings %d" % (
            it, loss.item(), lcg.stats["iterations"], float(lik.noise), float(model.covar_module.outputscale),
            [round(float(v), 3) for v in model.covar_module.base_kernel.lengthscale.reshape(-1)], rmse, lcg.stats["last_iterations"], len(w)), flush=True)
