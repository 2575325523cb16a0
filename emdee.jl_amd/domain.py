"""Spatial domain decomposition with ghost-atom halo exchange (new, the reference is
single-GPU).  One process per GPU; collectives go through torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

DomainPlan is device-agnostic host logic (pure torch): brick ownership, atom migration at rebuild,
ghost selection with periodic shifts, and the per-step position exchange as ONE all_to_all_single
(RCCL runs it as grouped send/recv: one message per xGMI peer, all links busy).  It never computes a
force.  DecomposedVerlet couples a plan to the HIP engine (VelocityVerlet's split step).

Geometry: the global box [0, L_d) is cut into g_x x g_y x g_z bricks.  Along a dimension with
g_d == 1 the rank keeps the whole periodic extent (the kernels wrap cells there); along a cut
dimension the rank's local box is its brick plus a halo of width h = cutoff + skin filled with ghost
atoms, which are images shifted by +-L_d where the neighbour wraps around the global box.  With a
full (owner-computes) neighbour list only positions travel; no force is sent back.
"""
import math

import numpy as np
import torch
import torch.distributed as dist


def rank_grid(world):
    """Most cubic factorisation of the rank count: 1->(1,1,1) 2->(2,1,1) 4->(2,2,1) 8->(2,2,2)."""
    best = None
    for gx in range(1, world + 1):
        if world % gx:
            continue
        for gy in range(1, world // gx + 1):
            if (world // gx) % gy:
                continue
            gz = world // (gx * gy)
            g = tuple(sorted((gx, gy, gz), reverse=True))
            score = max(g) / min(g)
            if best is None or score < best[0]:
                best = (score, g)
    return best[1]


class DomainPlan:
    def __init__(self, lengths, halo, world=None, rank=None, device="cpu", group=None, grid=None, transport="device"):
        self.group = group
        # "device": hand device tensors to the backend (RCCL moves HBM -> xGMI -> HBM directly);
        # "host": stage through pinned-less CPU copies, for backends that only take CPU tensors (gloo)
        self.transport = transport
        self.world = dist.get_world_size(group) if world is None else int(world)
        self.rank = dist.get_rank(group) if rank is None else int(rank)
        self.device = torch.device(device)
        self.L = [float(v) for v in lengths]
        self.halo = float(halo)
        self.grid = tuple(grid) if grid is not None else rank_grid(self.world)
        assert self.grid[0] * self.grid[1] * self.grid[2] == self.world
        gx, gy, gz = self.grid
        self.coords = (self.rank % gx, (self.rank // gx) % gy, self.rank // (gx * gy))
        self.width = [self.L[d] / self.grid[d] for d in range(3)]
        self.lo = [self.coords[d] * self.width[d] for d in range(3)]
        self.hi = [self.lo[d] + self.width[d] for d in range(3)]
        for d in range(3):
            if self.grid[d] > 1 and self.halo > self.width[d]:
                raise ValueError("halo %g exceeds the brick width %g along dimension %d" % (self.halo, self.width[d], d))
            if self.grid[d] == 1 and 2.0 * self.halo > self.L[d]:
                raise ValueError("cutoff + skin exceeds half the periodic length along dimension %d" % d)
        # local box handed to the engine: brick + halo on cut dimensions, whole period otherwise
        self.cut = [g > 1 for g in self.grid]
        self.local_lo = [self.lo[d] - self.halo if self.cut[d] else 0.0 for d in range(3)]
        self.local_len = [self.width[d] + 2 * self.halo if self.cut[d] else self.L[d] for d in range(3)]
        self.periodic = [0 if self.cut[d] else 1 for d in range(3)]
        # 26 directions, fixed order; only cut dimensions may be non-zero
        self.dirs = [(sx, sy, sz) for sz in (-1, 0, 1) for sy in (-1, 0, 1) for sx in (-1, 0, 1)
                     if (sx, sy, sz) != (0, 0, 0) and all(s == 0 or self.cut[d] for d, s in enumerate((sx, sy, sz)))]
        self.dir_rank, self.dir_shift = [], []
        for s in self.dirs:
            n, shift = [], []
            for d in range(3):
                c = self.coords[d] + s[d]
                shift.append(-self.L[d] if c >= self.grid[d] else (self.L[d] if c < 0 else 0.0))
                n.append(c % self.grid[d])
            self.dir_rank.append(n[0] + gx * (n[1] + gy * n[2]))
            self.dir_shift.append(shift)
        self.shift_table = torch.tensor(self.dir_shift if self.dirs else [[0.0, 0.0, 0.0]], dtype=torch.float64)
        self.send_ids = self.send_codes = None
        self.send_counts = self.recv_counts = None
        self.n_ghost = 0

    # ------------------------------------------------------------------ ownership
    def wrap(self, x):
        """Positions wrapped into the global periodic box."""
        L = torch.tensor(self.L, dtype=x.dtype, device=x.device)
        return x - L * torch.floor(x / L)

    def owner_of(self, xw):
        """Rank owning each (wrapped) position."""
        gx, gy, gz = self.grid
        c = []
        for d in range(3):
            cd = torch.floor(xw[:, d] / self.width[d]).to(torch.int64).clamp_(0, self.grid[d] - 1)
            c.append(cd)
        return c[0] + gx * (c[1] + gy * c[2])

    def _a2a(self, send, send_counts, recv_counts):
        """all_to_all_single of the rows of `send` with the given per-rank row counts."""
        shape = (int(sum(recv_counts)),) + tuple(send.shape[1:])
        if self.world == 1:
            return send.clone()
        if self.transport == "host" and send.device.type != "cpu":
            out = torch.empty(shape, dtype=send.dtype)
            dist.all_to_all_single(out, send.cpu(), output_split_sizes=list(recv_counts),
                                   input_split_sizes=list(send_counts), group=self.group)
            return out.to(send.device)
        out = torch.empty(shape, dtype=send.dtype, device=send.device)
        dist.all_to_all_single(out, send, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts),
                               group=self.group)
        return out

    def _counts(self, send_counts):
        """Every rank tells every other how many rows it will send."""
        t = torch.tensor(send_counts, dtype=torch.int64, device=self.device if self.transport == "device" else "cpu")
        if self.world == 1:
            return list(send_counts)
        r = torch.empty_like(t)
        dist.all_to_all_single(r, t, group=self.group)
        return r.tolist()

    def _all_to_all_rows(self, rows, dest):
        """Send each row to rank dest[row]; returns the rows received (ordered by source rank)."""
        order = torch.argsort(dest, stable=True)
        rows = rows[order].contiguous()
        sc = torch.bincount(dest, minlength=self.world).tolist()
        return self._a2a(rows, sc, self._counts(sc))

    def migrate(self, x, v, atoms, gid):
        """Re-assign atoms to the bricks that contain them (called at every rebuild).  x (n,3), v (n,3),
        atoms (n,2) float32, gid (n,) int64 -> the same arrays for the atoms this rank now owns, positions
        wrapped into the global box.  Only the atoms that left the brick travel (float64 rows: x, v, the two
        LJAtom floats, the global id); the others keep their order, arrivals are appended."""
        xw = self.wrap(x)
        if self.world > 1:
            dest = self.owner_of(xw)
            leave = dest != self.rank
            rows = torch.cat([xw[leave].to(torch.float64), v[leave].to(torch.float64), atoms[leave].to(torch.float64),
                              gid[leave].to(torch.float64).unsqueeze(1)], dim=1)
            rows = self._all_to_all_rows(rows, dest[leave])
            stay = ~leave
            xw = torch.cat([xw[stay], rows[:, 0:3].to(x.dtype)])
            v = torch.cat([v[stay], rows[:, 3:6].to(v.dtype)])
            atoms = torch.cat([atoms[stay], rows[:, 6:8].to(torch.float32)])
            gid = torch.cat([gid[stay], rows[:, 8].to(torch.int64)])
        return xw.contiguous(), v.contiguous(), atoms.contiguous(), gid.contiguous()

    # ------------------------------------------------------------------ ghosts
    def build_ghosts(self, x, atoms):
        """Choose the owned atoms every neighbour needs as ghosts, exchange them, and remember the send
        lists for the per-step position exchange.  Returns (ghost positions, ghost atoms)."""
        dev = x.device
        ids_per_rank = [[] for _ in range(self.world)]
        codes_per_rank = [[] for _ in range(self.world)]
        # one pass per cut dimension over all atoms, then the 26 directions only over the shell atoms
        near_lo, near_hi = {}, {}
        shell = None
        for d in range(3):
            if self.cut[d]:
                near_lo[d] = x[:, d] < self.lo[d] + self.halo
                near_hi[d] = x[:, d] >= self.hi[d] - self.halo
                either = near_lo[d] | near_hi[d]
                shell = either if shell is None else (shell | either)
        if shell is not None:
            sidx = torch.nonzero(shell, as_tuple=False).squeeze(1)
            lo_s = {d: near_lo[d][sidx] for d in near_lo}
            hi_s = {d: near_hi[d][sidx] for d in near_hi}
            sidx32 = sidx.to(torch.int32)
        for k, s in enumerate(self.dirs):
            mask = None
            for d in range(3):
                if s[d] != 0:
                    m = hi_s[d] if s[d] > 0 else lo_s[d]
                    mask = m if mask is None else (mask & m)
            ids = sidx32[mask]
            ids_per_rank[self.dir_rank[k]].append(ids)
            codes_per_rank[self.dir_rank[k]].append(torch.full_like(ids, k))
        empty = torch.empty(0, dtype=torch.int32, device=dev)
        per_rank_ids = [torch.cat(l) if l else empty for l in ids_per_rank]
        per_rank_codes = [torch.cat(l) if l else empty for l in codes_per_rank]
        self.send_ids = torch.cat(per_rank_ids).contiguous()
        self.send_codes = torch.cat(per_rank_codes).contiguous()
        self.send_counts = [int(t.shape[0]) for t in per_rank_ids]
        self.recv_counts = self._counts(self.send_counts)
        self.n_ghost = int(sum(self.recv_counts))
        ghost_x = self.exchange(self.pack_torch(x))
        a_send = atoms[self.send_ids.long()].contiguous()
        ghost_atoms = self._exchange_rows(a_send)
        return ghost_x, ghost_atoms

    def pack_torch(self, x):
        """Reference packer (host logic / CPU tests): positions of the send list plus their image shift.
        The GPU engine packs with the HIP kernel behind emdee_md_pack_positions instead."""
        shifts = self.shift_table.to(device=x.device, dtype=x.dtype)
        return (x[self.send_ids.long()] + shifts[self.send_codes.long()]).contiguous()

    def _exchange_rows(self, send):
        return self._a2a(send, self.send_counts, self.recv_counts)

    def exchange_begin(self, send_buf):
        """Start the halo exchange without waiting for it.  With the device transport the collective runs
        on the backend's own stream (it waits for the pack kernel, not for later work), so kernels enqueued
        afterwards overlap with it; exchange_end makes the current stream wait for the data."""
        if self.world == 1 or self.transport != "device":
            return None, self._exchange_rows(send_buf)
        out = torch.empty((self.n_ghost,) + tuple(send_buf.shape[1:]), dtype=send_buf.dtype, device=send_buf.device)
        work = dist.all_to_all_single(out, send_buf, output_split_sizes=list(self.recv_counts),
                                      input_split_sizes=list(self.send_counts), group=self.group, async_op=True)
        return work, out

    def exchange_end(self, handle):
        work, out = handle
        if work is not None:
            work.wait()
        return out

    def exchange(self, send_buf):
        """The per-step halo exchange: packed positions out, ghost positions in (ordered by source rank,
        matching the ghost slots n_owned .. n_owned + n_ghost - 1)."""
        return self._exchange_rows(send_buf)


def _pair_cells(cells, grid):
    return tuple(int(cells) * g for g in grid)


def lattice_block(cells, grid, coords, scaling="strong"):
    """(ncells, block_lo, block_n) of the fcc lattice cells the rank at `coords` of `grid` generates.
    strong: ONE cells^3 box cut into blocks whose sizes differ by at most one cell per dimension;
    weak: every rank a cells^3 block of a (g_x cells, g_y cells, g_z cells) lattice."""
    cells = int(cells)
    if scaling == "strong":
        ncells = (cells,) * 3
        lo = [(c * cells) // g for c, g in zip(coords, grid)]
        hi = [((c + 1) * cells) // g for c, g in zip(coords, grid)]
    else:
        ncells = _pair_cells(cells, grid)
        lo = [c * cells for c in coords]
        hi = [(c + 1) * cells for c in coords]
    return ncells, lo, [h - l for l, h in zip(lo, hi)]


class DecomposedVerlet:
    """Velocity-Verlet over a decomposed box: every rank integrates its brick with the HIP engine and
    refreshes its ghosts once per step.  Rebuild (migration + new ghost lists + neighbour list) when
    any rank reports a displacement above skin/2, or at a fixed cadence."""

    def __init__(self, pkg, plan, x, v, atoms, gid, model, skin=0.3, dtype=torch.float64):
        self.pkg, self.plan, self.model, self.skin, self.dtype = pkg, plan, model, float(skin), dtype
        self.gid = gid
        self.n_global = None
        self.md = None
        # overlap the halo exchange with interior bricks (only the device transport runs asynchronously)
        import os
        self.overlap = plan.transport == "device" and os.environ.get("EMDEE_DD_OVERLAP", "1") != "0"
        self.fused = os.environ.get("EMDEE_DD_FUSED", "1") != "0"
        self.since_build = 0
        self._langevin = None
        self._load(x, v, atoms, gid)

    @property
    def grid(self):
        return self.plan.grid

    @property
    def n_owned(self):
        return self.md.n_owned

    def _load(self, x, v, atoms, gid):
        plan = self.plan
        x, v, atoms, gid = plan.migrate(x, v, atoms, gid)
        gx, ga = plan.build_ghosts(x, atoms)
        pos = torch.cat([x, gx.to(x.dtype)]).to(self.dtype).contiguous()
        at = torch.cat([atoms, ga]).contiguous()
        vel = v.to(self.dtype).contiguous()
        self.gid = gid
        if self.md is None:
            self.md = self.pkg.VelocityVerlet(pos, vel, None, self.model, at, skin=self.skin, lo=plan.local_lo,
                                              lengths=plan.local_len, periodic=plan.periodic, n_ghost=plan.n_ghost)
        else:
            self.md.set_state_(pos, vel, at, None, plan.n_ghost)
        self.atoms = atoms
        self.since_build = 0
        self._shifts = [c for row in plan.shift_table.tolist() for c in row]
        if self._langevin is not None:                 # the engine forgets the id array with the old state
            self.md.set_langevin_ids_(self.gid)

    def set_langevin_(self, gamma, temperature, seed=0, first_step=0):
        """Langevin thermostat on every rank; the noise is keyed by GLOBAL atom ids, so the decomposed run
        draws the numbers the undivided run would."""
        self._langevin = (gamma, temperature, seed) if gamma > 0 else None
        self.md.set_langevin_(gamma, temperature, seed, first_step)
        self.md.set_langevin_ids_(self.gid if self._langevin is not None else None)

    def _with_halo(self, compute):
        """The per-step pattern: pack -> exchange (in flight on the collective's stream) || compute(phase 1:
        interior bricks, whose LDS tile holds no ghost cell) -> unpack -> compute(phase 2: boundary bricks).
        Without ghosts, or when the transport cannot run asynchronously, one compute(phase 0)."""
        plan = self.plan
        if plan.n_ghost == 0 and plan.send_ids.shape[0] == 0:
            return compute(0)
        buf = self.md.pack_positions(plan.send_ids, self._shifts, codes=plan.send_codes)
        if not self.overlap:
            recv = plan.exchange(buf)
            if plan.n_ghost:
                self.md.unpack_ghosts_(recv, 0)
            return compute(0)
        try:
            handle = plan.exchange_begin(buf)
        except (RuntimeError, TypeError):          # backend without async all_to_all: exchange in line from now on
            self.overlap = False
            handle = (None, plan.exchange(buf))
        compute(1)
        recv = plan.exchange_end(handle)
        if plan.n_ghost:
            self.md.unpack_ghosts_(recv, 0)
        return compute(2)

    def _forces_with_halo(self):
        self._with_halo(lambda phase: self.md.forces_(self.pkg.FORCES, phase=phase))

    def _fused_step_with_halo(self, dt):
        """One inner step: forces at the current positions (ghosts refreshed on the way) with the full kick
        and the drift fused into the same kernel.  Falls back to the split kernels if the engine declines."""
        if self.fused:
            state = {"ok": True}

            def compute(phase):
                if state["ok"] and not self.md.fused_step_(dt, 1.0, phase=phase):
                    state["ok"] = False
                if not state["ok"] and phase != 1:          # engine has no tiled kernels for this box
                    self.md.forces_(self.pkg.FORCES)
                    self.md.kick_drift_(dt, 1.0)
            self._with_halo(compute)
            self.fused = state["ok"]
        else:
            self._forces_with_halo()
            self.md.kick_drift_(dt, 1.0)

    def _any_rank(self, flag):
        if self.plan.world == 1:
            return flag
        host = self.plan.transport == "host"
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cpu" if host else self.plan.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.plan.group)
        return bool(t.item())

    def rebuild_(self):
        st = self.md.state(positions=True, velocities=True, forces=False)
        n = self.md.n_owned
        self._load(st["positions"][:n], st["velocities"], self.atoms, self.gid)

    def step_(self, nsteps, dt, rebuild_every=0):
        """nsteps velocity-Verlet steps.  x_1 = x_0 + dt (v_0 + dt/2 f_0); every inner step is then one fused
        kernel pass (force + full kick + drift: the closing half kick of a step rides on the opening half kick
        of the next); the last step ends with a plain force pass and the closing half kick."""
        nsteps = int(nsteps)
        if nsteps <= 0:
            return
        self.md.kick_drift_(dt, 0.5)
        for s in range(1, nsteps + 1):
            self.since_build += 1
            rb = (self.since_build >= rebuild_every) if rebuild_every > 0 else self._any_rank(self.md.needs_rebuild())
            if rb:
                self.rebuild_()                   # migrates, rebuilds ghosts and lists, evaluates forces
                if s < nsteps:
                    self.md.kick_drift_(dt, 1.0)
            elif s == nsteps:
                self._forces_with_halo()
            else:
                self._fused_step_with_halo(dt)
        self.md.kick_(dt)

    def totals(self):
        """Global (potential, kinetic, virial) sums."""
        t = torch.tensor(self.md.totals(), dtype=torch.float64,
                         device="cpu" if self.plan.transport == "host" else self.plan.device)
        if self.plan.world > 1:
            dist.all_reduce(t, group=self.plan.group)
        return tuple(t.tolist())

    def observables(self):
        """Global temperature and pressure (see VelocityVerlet.observables)."""
        ep, ek, vir = self.totals()
        n = self.n_global if self.n_global is not None else self._count_global()
        v = float(self.plan.L[0] * self.plan.L[1] * self.plan.L[2])
        return dict(potential=ep, kinetic=ek, virial=vir, temperature=2.0 * ek / max(3 * n - 3, 1),
                    pressure=(2.0 * ek + vir) / (3.0 * v), density=n / v)

    def _count_global(self):
        t = torch.tensor([self.md.n_owned], dtype=torch.int64, device="cpu" if self.plan.transport == "host" else self.plan.device)
        if self.plan.world > 1:
            dist.all_reduce(t, group=self.plan.group)
        self.n_global = int(t.item())
        return self.n_global

    def gather_state(self):
        """(gid, positions, velocities, forces) of this rank's owned atoms, caller order = ascending gid."""
        st = self.md.state()
        n = self.md.n_owned
        return self.gid, st["positions"][:n], st["velocities"], st["forces"]

    @classmethod
    def synthetic(cls, cells, world, rank, device, model, precision=torch.float64, skin=0.3, mixture=False,
                  temperature=1.0, pkg=None, group=None, transport="device", scaling="weak"):
        """Synthetic fcc box of SURVEY.md 8(d), generated in parallel: every rank generates (and initially owns) one
        block of lattice cells.  scaling = "strong": ONE cells^3-cell box whatever the rank count, cut into
        rank_grid(world) blocks (sizes differ by at most one lattice cell per dimension; migrate() then hands every
        atom to the brick that contains it).  scaling = "weak": every rank a cells^3-cell brick of a
        (g_x cells, g_y cells, g_z cells) lattice.  Velocities get the global centre-of-mass and temperature
        corrections through two small all-reduces."""
        if pkg is None:
            from __graft_entry__ import load_package
            pkg = load_package()
        syn = pkg.synthetic
        grid = rank_grid(world)
        coords = (rank % grid[0], (rank // grid[0]) % grid[1], rank // (grid[0] * grid[1]))
        ncells, block_lo, block_n = lattice_block(cells, grid, coords, scaling)
        pos, gid, lengths = syn.fcc_block(ncells, block_lo, block_n)
        n_global = 4 * ncells[0] * ncells[1] * ncells[2]
        vel = syn.raw_normals(gid, n_global)
        if mixture:
            eps, sigma = syn.mixture_parameters(syn.mixture_types(gid))
            atoms = pkg.lennard_jones_atoms(eps, sigma)
        else:
            atoms = pkg.lennard_jones_atoms(1.0, 1.0, pos.shape[0])
        cdev = device if transport == "device" else "cpu"
        sums = torch.tensor(np.concatenate([vel.sum(axis=0), [0.0]]), dtype=torch.float64, device=cdev)
        if world > 1:
            dist.all_reduce(sums, group=group)
        vel = vel - (sums[:3].cpu().numpy() / n_global)
        ke = torch.tensor([0.5 * float(np.sum(vel * vel))], dtype=torch.float64, device=cdev)
        if world > 1:
            dist.all_reduce(ke, group=group)
        vel *= math.sqrt(temperature * max(3 * n_global - 3, 1) / (2.0 * ke.item()))
        rc = math.sqrt(model.rc2)
        plan = DomainPlan(lengths, rc + skin, world=world, rank=rank, device=device, group=group, grid=grid,
                          transport=transport)
        x = torch.from_numpy(pos).to(device)
        v = torch.from_numpy(vel).to(device)
        a = pkg.cu(atoms, device)
        g = torch.from_numpy(gid).to(device)
        self = cls(pkg, plan, x, v, a, g, model, skin=skin, dtype=precision)
        self.n_global = n_global
        return self
