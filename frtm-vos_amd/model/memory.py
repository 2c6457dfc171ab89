"""Sample memory of the target model (API of the reference's model/memory.py:4-92).

MI355X layout.  The reference keeps, per slot, the projected features AND the full-resolution
label and pixel-weight maps (2 x 1.64 MB per slot at 480p) and streams them through every CG
iteration.  Here each slot keeps the features plus the *low-resolution normal equations* of its
label/pixel-weight pair:

    normal_B[slot] (9,h,w) = U^T diag(pw^2) U  as a 3x3 stencil,     normal_c[slot] (h,w) = U^T (pw^2 * label)

built once per insert by one HIP kernel (``frtm_normal_build``); the CG loop never touches a
full-resolution tensor.  Sample weights live on the device and the replacement index never
comes back to the host (``frtm_memory_next_slot``): no ``.item()`` sync (reference memory.py:80-81).

``keep_hires=True`` additionally stores ``labels`` / ``pixel_weights`` like the reference
(debugging / API completeness; not read by the solver); without it reading the two attributes raises.
"""
import torch

from .. import _hip as H


class Memory:
    _init_weights = {}           # (K, device) -> the initial sample weights (2, 1, ..., 1) / (K + 1), formed once in the reference's arithmetic

    def __init__(self, capacity, feature_size, labels_size, device, learning_rates, grid_size=None,
                 pixel_weighting=None, keep_hires=False):
        """
        :param capacity:        number of slots
        :param feature_size:    (C,h,w) of a stored feature map
        :param labels_size:     (1,H,W) of the label image
        :param learning_rates:  sample-weight learning rate (0.1 in evaluate.py:34)
        :param grid_size:       (h,w) of the score grid the labels are compared on (default: feature grid)
        :param pixel_weighting: dict(method='hinge', tf=...) or None; used when update()/initialize() are
                                called without an explicit pixel-weight tensor
        """
        dev = torch.device(device)
        if dev.type != 'cuda':
            raise RuntimeError('Memory lives on the GPU (got device %s); there is no CPU path' % device)
        self.samples = torch.zeros(capacity, *feature_size, device=dev)
        self.weights = torch.zeros(capacity, device=dev)
        self.grid = tuple(grid_size) if grid_size is not None else tuple(feature_size[-2:])
        self.labels_size = tuple(labels_size)
        self.normal_B = torch.zeros(capacity, 9, *self.grid, device=dev)
        self.normal_c = torch.zeros(capacity, *self.grid, device=dev)
        self.keep_hires = False
        self._labels = self._pixel_weights = None
        self.pw_params = pixel_weighting
        self._capacity = capacity
        self.current_size = 0
        self.device = dev
        self.learning_rates = learning_rates
        # {previous_replace_ind, last index, inserts performed, inserts skipped by the device-side early-out}; fills, not blocking H2D copies
        self._slot = torch.zeros(4, dtype=torch.int32, device=dev)
        self._slots_w = None
        self._slot[:2].fill_(-1)
        self._have_prev = False
        self._scratch = torch.zeros(max(capacity, 8) * 32, device=dev)
        if keep_hires:
            self._alloc_hires()

    def _alloc_hires(self):
        self._labels = torch.zeros(self._capacity, *self.labels_size, device=self.device)
        self._pixel_weights = torch.zeros(self._capacity, *self.labels_size, device=self.device)
        self.keep_hires = True

    @property
    def labels(self):
        """Full-resolution label maps like the reference's Memory.labels (memory.py:15).  The solver only needs their low-resolution
        normal form, so they exist only in a memory constructed with ``keep_hires=True`` (Discriminator(..., keep_hires=True)).
        Reading the attribute never changes the memory's state (round-2 ADVICE: a debug read used to switch recording on, which
        silently took the window-insert path away and re-allocated the memory on recycle)."""
        if self._labels is None:
            raise AttributeError('Memory.labels: the full-resolution maps are not recorded (only their low-resolution normal equations '
                                 'normal_B / normal_c are); construct the memory with keep_hires=True to keep them')
        return self._labels

    @property
    def pixel_weights(self):
        """Full-resolution pixel-weight maps (reference memory.py:16); like ``labels`` only with keep_hires=True."""
        if self._pixel_weights is None:
            raise AttributeError('Memory.pixel_weights: not recorded; construct the memory with keep_hires=True')
        return self._pixel_weights

    @classmethod
    def aliasing(cls, samples, labels, pixel_weights, weights, learning_rates=0.1):
        """A memory that ALIASES the caller's buffers (the reference's problem objects alias memory.samples / labels /
        pixel_weights / weights, discriminator.py:169-171,188-191: in-place edits between two run() calls reach the solver).
        Only the low-resolution normal equations are own storage; ``refresh_normals`` rebuilds them from the aliased maps."""
        m = cls.__new__(cls)
        dev = samples.device
        if dev.type != 'cuda':
            raise RuntimeError('Memory lives on the GPU (got device %s); there is no CPU path' % dev)
        cap = samples.shape[0]
        m.samples, m.weights = samples, weights
        m.grid = tuple(samples.shape[-2:])
        m.labels_size = tuple(labels.shape[1:])
        m.normal_B = torch.zeros(cap, 9, *m.grid, device=dev)
        m.normal_c = torch.zeros(cap, *m.grid, device=dev)
        m.keep_hires = True
        m._labels, m._pixel_weights = labels, pixel_weights
        m.pw_params = None
        m._capacity = cap
        m.current_size = 0
        m.device = dev
        m.learning_rates = learning_rates
        m._slot = torch.zeros(4, dtype=torch.int32, device=dev)
        m._slots_w = None
        m._slot[:2].fill_(-1)
        m._have_prev = False
        m._scratch = torch.zeros(max(cap, 8) * 32, device=dev)
        return m

    def refresh_normals(self):
        """Active samples = weight > 0 (reference discriminator.py:38-43); they must be the leading slots (slots fill in index
        order).  Rebuilds normal_B / normal_c of those slots from the full-resolution maps and returns their number."""
        active = (self.weights > 0).nonzero().flatten().tolist()
        n = len(active)
        if active != list(range(n)):
            raise ValueError('active samples (weight > 0) must occupy the leading memory slots, got %s' % active)
        if n:
            self._build_normals(self._labels[:n], self._pixel_weights[:n], n, None, 0)
        self.current_size = n
        return n

    @property
    def capacity(self):
        return self._capacity

    @property
    def previous_replace_ind(self):
        """Host view of the last replaced slot (synchronises; the hot path never reads it)."""
        return int(self._slot[0].item()) if self._have_prev else None

    @property
    def insert_counts(self):
        """(inserts performed, inserts skipped by the device-side "fewer than 10 pixels" guard) since construction / reset();
        counted by the slot kernel itself.  Synchronises: diagnostics (bench.py, tests), never read on the hot path."""
        a, b = self._slot[2:].tolist()
        return int(a), int(b)

    def _tf(self):
        p = self.pw_params
        if p is None or p.get('method', 'none') == 'none':
            return -1.0
        assert p['method'] == 'hinge'
        return float(p['tf'])

    def clear(self):
        self.current_size = 0
        H.fill(self.weights, 0.0)

    def reset(self):
        """Back to the state after construction (buffers are kept: a recycled memory serves the next object)."""
        self.clear()
        H.fill(self._slot[:2], -1)
        H.fill(self._slot[2:], 0)
        self._have_prev = False

    def matches(self, capacity, feature_size, labels_size, grid_size=None, keep_hires=False):
        grid = tuple(grid_size) if grid_size is not None else tuple(feature_size[-2:])
        return (self._capacity == capacity and tuple(self.samples.shape[1:]) == tuple(feature_size) and
                self.labels_size == tuple(labels_size) and self.grid == grid and self.keep_hires == keep_hires)

    def _build_normals(self, labels, pixel_weights, n, slot_dev, slot_host, px_count=None):
        Hh, Ww = self.labels_size[-2:]
        lab = labels.reshape(n, Hh, Ww)
        if lab.dtype != torch.uint8:
            lab = lab.float()
        lab = lab.contiguous()
        pw = None if pixel_weights is None else pixel_weights.reshape(n, Hh, Ww).float().contiguous()
        H.call('frtm_normal_build', H.ptr(lab), int(lab.dtype == torch.uint8), H.ptr(pw), n, Hh, Ww,
               self.grid[0], self.grid[1], self._tf(), slot_dev, slot_host,
               H.ptr(self.normal_B), H.ptr(self.normal_c), H.ptr(self._scratch), None if px_count is None else px_count.data_ptr())
        return lab, pw

    def initialize(self, init_features, init_labels, pixel_weights=None):
        """Reference memory.py:33-48.  pixel_weights=None: hinge weights are computed inside the kernel."""
        K = init_features.shape[0]
        assert init_labels.shape[0] == K and K <= self._capacity
        self.samples[:K].copy_(init_features.detach())
        # (2, 1, ..., 1) / (K + 1) in the reference's float32 arithmetic (memory.py:38-46: w = [2/K, 1/K, ...]; w / w.sum()), computed once per
        # K and kept on the device: the per-sequence path is one device-to-device copy, no framework kernel
        cache = Memory._init_weights.get((K, str(self.device)))
        if cache is None:
            w = torch.full((K,), 1.0 / K, device=self.device)
            w[:1].fill_(2.0 / K)
            cache = (w / w.sum()).contiguous()
            # formed ONCE per process on whatever stream this object's fit runs on, read by every later fit on ITS stream: objects that start
            # together are fitted on concurrent streams, and the second one must not read the table before these kernels have run (it did:
            # non-finite filters for the second object whenever recycled device memory held something else than zeros)
            torch.cuda.current_stream(cache.device).synchronize()
            Memory._init_weights[(K, str(self.device))] = cache
        self.weights[:K].copy_(cache)
        lab, pw = self._build_normals(init_labels, pixel_weights, K, None, 0)
        if self.keep_hires:
            self._labels[:K] = lab.float().view(K, *self.labels_size)
            self._pixel_weights[:K] = (pw if pw is not None else self._hires_pw(lab)).view(K, *self.labels_size)
        self.current_size = K

    def initialize_like(self, init_features, other):
        """initialize() with the labels / pixel weights of ``other`` (a memory initialised from the same label images on the same
        grid): their low-resolution normal equations are copied instead of being rebuilt from the full-resolution labels.  Only
        static device buffers are touched, so the call can be part of a captured hipGraph (Discriminator.init)."""
        K = init_features.shape[0]
        assert K == other.current_size and K <= self._capacity and self.grid == other.grid and not self.keep_hires
        self.samples[:K].copy_(init_features.detach())
        self.weights[:K].copy_(other.weights[:K])     # (2, 1, ..., 1) / (K + 1): memory.py:38-46, same for both memories
        self.normal_B[:K].copy_(other.normal_B[:K])
        self.normal_c[:K].copy_(other.normal_c[:K])
        self.current_size = K

    def _hires_pw(self, lab):
        n = lab.shape[0]
        out = torch.empty(lab.shape, device=self.device, dtype=torch.float32)
        ys = (lab.float() > 0.5).float().contiguous()
        H.call('frtm_pixel_weights', H.ptr(ys), 0, n, lab.shape[-2], lab.shape[-1], self._tf(), H.ptr(out), H.ptr(self._scratch))
        return out

    def update_sample_weights(self, previous_replace_ind=None, count_dev=None, min_count=10):
        """Reference memory.py:65-92, on the device.  Returns nothing: the chosen slot stays in ``self._slot``.
        ``count_dev``: optional device int32 (pixels > 0.5); below ``min_count`` the whole update becomes a no-op on the
        device (slot -1), which is how Discriminator.update's early-out runs without a host sync."""
        H.call('frtm_memory_next_slot', H.ptr(self.weights), self._capacity, float(self.learning_rates),
               int(self.current_size == 0), H.ptr(self._slot), None if count_dev is None else count_dev.data_ptr(), int(min_count))
        self._have_prev = True

    def insert_at(self, slot_dev_ptr, ft, labels, pixel_weights, px_count=None):
        """Reference memory.py:50-57; the slot is a device-resident index."""
        ft = ft.detach().contiguous()
        H.call('frtm_memory_insert', H.ptr(ft), H.ptr(self.samples), ft.numel(), slot_dev_ptr)
        lab, pw = self._build_normals(labels, pixel_weights, 1, slot_dev_ptr, 0, px_count if pixel_weights is None else None)
        if self.keep_hires:
            labf = lab.float().contiguous()          # named: H.ptr() only takes the address
            H.call('frtm_memory_insert', H.ptr(labf), H.ptr(self._labels), labf.numel(), slot_dev_ptr)
            pwt = pw if pw is not None else self._hires_pw(lab)
            H.call('frtm_memory_insert', H.ptr(pwt), H.ptr(self._pixel_weights), pwt.numel(), slot_dev_ptr)

    def update_window(self, features, masks, plane, counts):
        """update() for W consecutive frames in three launches (frtm_memory_update_window) instead of 3 W: ``features`` (W,c,h,w)
        dense, the soft label of frame f is ``masks[f, plane]`` (masks: (W,K,H,W) float, dense), its pixel count
        ``counts[f, plane]`` (int32 (W,K), from ops.count_above); frames with fewer than 10 pixels are skipped on the device like
        update(count_dev=...) does.  Same slots, weights, samples and normal equations as W update() calls (no full-resolution
        copies: ``keep_hires`` memories take the frame-by-frame path)."""
        W, K = masks.shape[0], masks.shape[1]
        Hh, Ww = self.labels_size[-2:]
        assert not self.keep_hires
        assert features.is_contiguous() and masks.is_contiguous() and masks.dtype == torch.float32 and counts.is_contiguous()
        assert features.shape[0] == W and tuple(masks.shape[-2:]) == (Hh, Ww) and tuple(counts.shape) == (W, K) and counts.dtype == torch.int32
        if self._slots_w is None or self._slots_w.numel() < W:
            self._slots_w = torch.empty(max(64, W), dtype=torch.int32, device=self.device)
        ln = features[0].numel()
        H.call('frtm_memory_update_window', H.ptr(self.weights), self._capacity, float(self.learning_rates), int(self.current_size == 0),
               H.ptr(self._slot), counts.data_ptr() + 4 * plane, K, 10, W, H.ptr(self._slots_w), H.ptr(features), H.ptr(self.samples), ln,
               masks.data_ptr() + 4 * plane * Hh * Ww, K * Hh * Ww, Hh, Ww, self.grid[0], self.grid[1], self._tf(),
               H.ptr(self.normal_B), H.ptr(self.normal_c), H.ptr(self._scratch))
        self._have_prev = True
        self.current_size = min(self.current_size + W, self._capacity)

    def update(self, features, labels, pixel_weights=None, count_dev=None, px_count=None):
        """Reference memory.py:59-63.  With ``count_dev`` the insert is guarded on the device; ``current_size`` is then an
        upper bound (a skipped insert leaves a zero-weight slot, which contributes nothing to the solver).  ``px_count``
        (default: ``count_dev``): device int32 with the number of label pixels > 0.5, if the caller already has it."""
        self.update_sample_weights(count_dev=count_dev)
        self.insert_at(self._slot[1:].data_ptr(), features, labels, pixel_weights, px_count if px_count is not None else count_dev)
        self.current_size = min(self.current_size + 1, self._capacity)
