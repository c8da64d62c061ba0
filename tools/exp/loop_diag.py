import sys, time, torch
sys.path.insert(0, '.')
from deformcontact_amd import dp
from deformcontact_amd.graphnet import EVERYDAY_NETWORK, load_model
from deformcontact_amd.loaders import InMemoryDataset, PrefetchLoader, SyntheticEverydayDataset
from deformcontact_amd.train import GraphedTrainStep
dev = torch.device('cuda:0')
ds = InMemoryDataset(SyntheticEverydayDataset(48 * 4))
t = time.perf_counter(); n = 0
for _, b in PrefetchLoader(ds, 4, dev, shuffle=False, depth=3):
    n += 1
torch.cuda.synchronize()
print('loader only (pin + upload): %.2f ms/batch' % ((time.perf_counter() - t) / n * 1e3))
t = time.perf_counter(); n = 0
for _, b in PrefetchLoader(ds, 4, None, shuffle=False, depth=3):
    n += 1
print('loader only (host): %.2f ms/batch' % ((time.perf_counter() - t) / n * 1e3))
torch.manual_seed(0)
model = load_model(EVERYDAY_NETWORK).to(dev)
bucket = dp.GradBucket(model.parameters(), direct=True)
opt = dp.FlatAdam(bucket, lr=4e-4, zero_grad_in_step=True)
bucket.zero()
stepper = GraphedTrainStep(model, opt, bucket, 1.0, eager_steps=2)
batches = [b for _, b in PrefetchLoader(ds, 4, dev, shuffle=False, depth=3)][:8]
for b in batches[:4]:
    stepper(*b)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(40):
    stepper(*batches[i % 8])
torch.cuda.synchronize()
print('stepper only (resident batches): %.2f ms/step' % ((time.perf_counter() - t) / 40 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(20):
    stepper(*batches[i % 8])
torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(8)
