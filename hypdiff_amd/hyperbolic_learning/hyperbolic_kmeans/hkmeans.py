"""Hyperbolic k-means on HIP kernels -- drop-in for `HyperbolicKMeans` of the reference's
hyperbolic_learning/hyperbolic_kmeans/hkmeans.py (the step between `lp_train` and `diff.hyperdiff`
that produces label.npy / center.npy, hyperkmeans.py:21-58).

Same constructor, `fit(X, y=None, max_epochs=40)` -> (centroids, labels) as numpy arrays, and the
attributes `centroids`, `assignments` (one-hot), `labels`, `inertia_`.  The reference loops in
Python over samples x clusters x gradient rounds; here one epoch is 10 x (one partial-sum kernel
+ one centre-update kernel) and one assignment kernel, all fp64 like numpy.  The two random
initialisations use numpy's global generator with the reference's call sequence
(`np.random.uniform(0, radius, (K, dim))`, then one `np.random.randint(0, K)` per sample), so a
seeded run starts exactly where the reference does; they can also be passed in."""
import numpy as np
import torch

from ... import _lib, ops


class HyperbolicKMeans():
    def __init__(self, n_clusters=6, max_iter=300, tol=1e-8, verbose=False):
        self.n_clusters = n_clusters
        self.max_iter = max_iter
        self.tol = tol
        self.verbose = verbose
        self.labels = None
        self.cluster_centers_ = None
        self.device = "cuda"

    def init_centroids(self, dim=None, radius=0.3):
        """hkmeans.py:97-111"""
        self.centroids = np.random.uniform(0, radius, size=(self.n_clusters, dim))

    def init_assign(self, labels=None):
        """hkmeans.py:113-125 (as a label vector; `assignments` is rebuilt one-hot at the end)"""
        if labels is not None:
            self._lab = np.asarray(labels, dtype=np.int32).copy()
        else:
            self._lab = np.array([np.random.randint(0, self.n_clusters) for _ in range(self.n_samples)], dtype=np.int32)

    # ---- device steps ------------------------------------------------------------------
    def _to_h(self, y):
        h = torch.empty(y.shape[0], y.shape[1] + 1, dtype=torch.float64, device=y.device)
        _lib.call("hd_hkm_to_hyperboloid", ops._p(y), ops._p(h), y.shape[0], y.shape[1], ops._stream())
        return h

    def _update_centroids(self, H, lab, cen, ws, rounds=10, alpha=0.1):
        """update_centroids hkmeans.py:127-150: compute_mean(theta_k, H_k, alpha=0.1), 10 rounds"""
        theta = self._to_h(cen)
        n, d = H.shape[0], H.shape[1] - 1
        for _ in range(rounds):
            _lib.call("hd_hkm_mean_step", ops._p(H), ops._p(lab), ops._p(theta), n, d, self.n_clusters, alpha, ops._p(ws),
                      ops._stream())
        out = torch.empty_like(cen)
        _lib.call("hd_hkm_to_poincare", ops._p(theta), ops._p(out), self.n_clusters, d, ops._stream())
        # a cluster without members keeps its Poincare centre exactly (hkmeans.py:138-139)
        counts = torch.bincount(lab.to(torch.int64), minlength=self.n_clusters)
        return torch.where((counts == 0).unsqueeze(1), cen, out)

    def _assign(self, X, cen, lab, d2):
        _lib.call("hd_hkm_assign", ops._p(X), ops._p(cen), ops._p(lab), ops._p(d2), X.shape[0], X.shape[1],
                  self.n_clusters, ops._stream())

    def fit(self, X, y=None, max_epochs=40, verbose=False, init_centroids=None, init_labels=None):
        """hkmeans.py:161-217"""
        Xd = torch.as_tensor(np.asarray(X) if not torch.is_tensor(X) else X).to(device=self.device, dtype=torch.float64)
        Xd = Xd.contiguous()
        nrm = torch.linalg.norm(Xd, dim=1)
        if bool((nrm > 1).any()):                      # :184-185
            Xd = (Xd / nrm.max()).contiguous()
        self.n_samples = Xd.shape[0]
        if init_centroids is None:
            self.init_centroids(dim=Xd.shape[1])
        else:
            self.centroids = np.asarray(init_centroids, dtype=np.float64)
        if init_labels is not None:
            self._lab = np.asarray(init_labels, dtype=np.int32).copy()
        else:
            self.init_assign(y)
        cen = torch.from_numpy(np.ascontiguousarray(self.centroids, dtype=np.float64)).to(self.device)
        lab = torch.from_numpy(self._lab).to(self.device)
        H = self._to_h(Xd)
        ws = torch.empty(int(_lib.load().hd_hkm_workspace_doubles(Xd.shape[0], Xd.shape[1], self.n_clusters)),
                         dtype=torch.float64, device=self.device)
        d2 = torch.empty(Xd.shape[0], dtype=torch.float64, device=self.device)
        if y is not None:
            cen = self._update_centroids(H, lab, cen, ws)          # :193-195
        for _ in range(max_epochs):
            cen = self._update_centroids(H, lab, cen, ws)
            self._assign(Xd, cen, lab, d2)
        self.inertia_ = float(d2.sum()) if max_epochs > 0 else 0
        self.centroids = cen.cpu().numpy()
        self.labels = lab.cpu().numpy().astype(np.int64)
        self.assignments = np.zeros((self.n_samples, self.n_clusters))
        self.assignments[np.arange(self.n_samples), self.labels] = 1
        self.centroids_device, self.labels_device = cen, lab
        return self.centroids, self.labels

    def predict(self, X):
        """hkmeans.py:219-235 -> one-hot (n, K)"""
        Xd = torch.as_tensor(np.asarray(X) if not torch.is_tensor(X) else X).to(device=self.device, dtype=torch.float64)
        cen = torch.from_numpy(np.ascontiguousarray(self.centroids, dtype=np.float64)).to(self.device)
        lab = torch.empty(Xd.shape[0], dtype=torch.int32, device=self.device)
        self._assign(Xd.contiguous(), cen, lab, None)
        out = np.zeros((Xd.shape[0], self.n_clusters))
        out[np.arange(Xd.shape[0]), lab.cpu().numpy()] = 1
        return out
