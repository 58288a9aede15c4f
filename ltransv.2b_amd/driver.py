"""External / internal time-step loops around the engine.

Mirror of run_LTRANS (LTRANS.f90:141-161), run_External_Timestep (:548-579) and
run_Internal_Timestep (:583-612): the host rotates the three hydrodynamic time
slots and uploads ONE new record per external step exactly where the reference
calls updateHydro (:559, hydrodynamic_module.f90:1052-1410), and calls
``engine.update_particles(ex, ix, p, it)`` exactly where the reference calls
update_particles() (:596).
"""
import numpy as np


class LtransRun:
    def __init__(self, engine, dt, idt, record_fn, rom_fn=None, dims=None, window_fn=None):
        """record_fn(r) -> dict(zeta, Uvel, Vvel, Wvel, Kh) for history record r (0-based).

        rom_fn(r) -> dict(romZ, romU, romV, romW, romKH) (the record as read from the file,
        unmasked) switches updateHydro to the slab pipeline: the record of the NEXT external
        step is handed to ``slab_prefetch`` as soon as the current one has been committed,
        and ``update_hydro`` is only the rotation + ``slab_commit``.  dims = (vi,uj,ui,vj);
        window_fn(r) -> t_ijruv(12) or None."""
        self.e = engine
        self.dt, self.idt = int(dt), int(idt)
        self.record_fn = record_fn
        self.rom_fn, self.dims, self.window_fn = rom_fn, dims, window_fn
        self._prefetched = None
        self.t_b, self.t_c, self.t_f = 1, 2, 3
        self.stepf = 3
        self.p = 0
        self.steps_done = 0

    def init_hydro(self):
        """initHydro (hydrodynamic_module.f90:661-1049): first three records."""
        for slot in (1, 2, 3):
            r = self.record_fn(slot - 1)
            self.e.set_slab(slot, r["zeta"], r["Uvel"], r["Vvel"], r["Wvel"], r.get("Kh"),
                            r.get("salt"), r.get("temp"))
        self.t_b, self.t_c, self.t_f = 1, 2, 3
        self.stepf = 3
        self.e.set_time_slots(1, 2, 3)

    def update_hydro(self):
        """updateHydro: rotate indices (:1080-1082), read one record into slot t_f."""
        self.t_b = self.t_b % 3 + 1
        self.t_c = self.t_c % 3 + 1
        self.t_f = self.t_f % 3 + 1
        self.stepf += 1
        if self.rom_fn is not None:
            if self._prefetched != self.stepf - 1:
                self.prefetch(self.stepf - 1)
            self.e.slab_commit(self.t_b, self.t_c, self.t_f)
            self._prefetched = None
            return
        r = self.record_fn(self.stepf - 1)
        self.e.set_slab(self.t_f, r["zeta"], r["Uvel"], r["Vvel"], r["Wvel"], r.get("Kh"),
                        r.get("salt"), r.get("temp"))
        self.e.set_time_slots(self.t_b, self.t_c, self.t_f)

    def prefetch(self, r):
        """hand history record r (0-based) to the engine's spare slot (asynchronous)"""
        w = self.window_fn(r) if self.window_fn is not None else None
        self.e.slab_prefetch(self.rom_fn(r), self.dims, masked=False, window=w)
        self._prefetched = r

    def prefetch_next(self):
        """the record the NEXT update_hydro will need; call right after update_hydro (or
        init_hydro) so that the upload runs beside this external step's internal steps"""
        if self.rom_fn is not None and self._prefetched is None:
            self.prefetch(self.stepf)

    def external_times(self, p):
        return np.array([(p - 2) * self.dt, (p - 1) * self.dt, p * self.dt], dtype=np.float64)

    def internal_times(self, ex, it):
        return np.array([ex[1] + float((it - 2) * self.idt), ex[1] + float((it - 1) * self.idt),
                         ex[1] + float(it * self.idt)], dtype=np.float64)

    def run_external_timestep(self, p, max_internal=None, on_step=None):
        self.p = p
        if p > 2:
            self.update_hydro()
        if p >= 2:
            self.prefetch_next()
        ex = self.external_times(p)
        stepIT = self.dt // self.idt
        if max_internal is not None:
            stepIT = min(stepIT, max_internal)
        for it in range(1, stepIT + 1):
            ix = self.internal_times(ex, it)
            self.e.update_particles(ex, ix, p, it)
            self.steps_done += 1
            if on_step is not None:
                on_step(p, it, ex, ix)

    def run(self, n_external, max_internal=None, on_step=None):
        for p in range(1, n_external + 1):
            self.run_external_timestep(p, max_internal, on_step)
