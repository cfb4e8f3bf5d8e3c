"""Host-side contact matrix: mobility factors, mask probabilities and the sampling tables.

Mirror of the reference's `ContactMatrix` (cythonsim/main.pyx:1119-1288): same methods
(`set_mobility_factor`, `set_mask_probability`, `init_day`), same table semantics, numpy instead
of pandas.  The tables it produces are what the device kernels (and the CPU oracles) sample from:

  * ``nr_contacts_by_age[a]``  float64 — total contacts/day of a participant aged ``a`` after the
    multiplicative mobility factors (main.pyx:1198-1211);
  * per age, the list of (place, contact_age_min, contact_age_max, cum_p, mask_p) entries in the
    reference's order — sorted by place-type STRING then contact-age tuple (main.pyx:1213-1235,
    pandas `sort_index`) — with ``cum_p`` the running sum of contacts/total.

Bit-compat notes (checked end-to-end by the sequential oracle against the cythonsim goldens):
  * per-age totals follow pandas' groupby-sum, which is a Kahan-compensated sum in row order;
  * mobility factors and the scalar `mobility_factor` are stored as C floats in the reference
    (`cdef public float`, main.pyx:1110,1128), i.e. rounded to float32 before use;
  * `wear-masks` only edits the mask matrix; tables pick it up at the next rebuild (quirk Q4).
"""
import numpy as np

PLACES = ('home', 'work', 'school', 'transport', 'leisure', 'other')  # enum order main.pyx:64-70
PLACE_ALL = 100
_parsed_cache = {}


class ContactTables:
    """Flat arrays for one rebuild. Entries of age a are [offset[a], offset[a]+count[a])."""

    def __init__(self, nr_contacts_by_age, offset, count, place, cmin, cmax, cum_p, mask_p):
        self.nr_contacts_by_age = nr_contacts_by_age
        self.offset = offset
        self.count = count
        self.place = place
        self.cmin = cmin
        self.cmax = cmax
        self.cum_p = cum_p
        self._mask_p = mask_p
        self.mask_source = None

    @property
    def mask_p(self):
        if self._mask_p is None and self.mask_source is not None:
            probs, ages, places = self.mask_source
            self._mask_p = probs[ages, places].astype(np.float32)
        return self._mask_p


class ContactMatrix:
    def __init__(self, contacts_per_day, nr_ages):
        """contacts_per_day: iterable of (place_type, participant_age, (cmin, cmax), contacts) or a
        DataFrame with those columns (calc/simulation.py:74-100)."""
        self.nr_ages = nr_ages
        key = (id(contacts_per_day), nr_ages)
        parsed = _parsed_cache.get(key)
        if parsed is None or parsed[0] is not contacts_per_day:
            parsed = (contacts_per_day, self._parse(contacts_per_day, nr_ages))
            _parsed_cache.clear()       # one scenario at a time; keeps the source object alive
            _parsed_cache[key] = parsed
        (self._place, self._page, self._cmin, self._cmax, self._contacts, self._rank, self._rows_of_age,
         self._sorted_rows, self._uniform, self._rows_mat, self._sorted_mat) = parsed[1]
        # host-side builder of the library (engine.f['build_contact_tables']) and the packed table shape;
        # set by the Context that owns the matrix, None = the numpy form below
        self.native_build = None
        self._native_static = None
        self.pack_ages, self.pack_entries = 0, 0
        self.mobility_factors = []  # [place, min_age, max_age, factor(float32)]
        self.mobility_factor = np.float32(1.0)
        self.mobility_factor_changed = False
        self.mask_probabilities = np.zeros((nr_ages, len(PLACES)), dtype=np.float64)
        # the tables before any mobility factor depend only on the rows: built once per parsed set
        if len(parsed) < 3:
            parsed = parsed + (self.generate_contact_probabilities(),)
            _parsed_cache[key] = parsed
        self.tables = parsed[2]

    @staticmethod
    def _parse(contacts_per_day, nr_ages):
        """Row arrays and the (static) entry order; shared by every ContactMatrix built from the same
        rows object (ensembles construct many)."""
        if hasattr(contacts_per_day, 'itertuples'):
            rows = [(t.place_type, int(t.participant_age), tuple(t.contact_age), float(t.contacts))
                    for t in contacts_per_day.itertuples()]
        else:
            rows = [(r[0], int(r[1]), tuple(r[2]), float(r[3])) for r in contacts_per_day]
        place = np.array([PLACES.index(r[0]) for r in rows], dtype=np.int32)
        page = np.array([r[1] for r in rows], dtype=np.int32)
        cmin = np.array([r[2][0] for r in rows], dtype=np.int32)
        cmax = np.array([r[2][1] for r in rows], dtype=np.int32)
        contacts = np.array([r[3] for r in rows], dtype=np.float64)
        # sort key of the reference's MultiIndex: (place_type string, contact_age tuple)
        place_rank = {p: i for i, p in enumerate(sorted(PLACES))}
        rank = np.array([place_rank[r[0]] for r in rows], dtype=np.int64)
        # per-age row lists in original order, and the reference's entry order per age
        # (sort_index on (place_type string, contact_age tuple); depends only on the keys)
        rows_of_age = [np.nonzero(page == a)[0] for a in range(nr_ages)]
        sorted_rows = []
        for a in range(nr_ages):
            r = rows_of_age[a]
            order = np.lexsort((cmax[r], cmin[r], rank[r]))
            sorted_rows.append(r[order])
        counts = {len(r) for r in rows_of_age}
        uniform = len(counts) == 1 and counts != {0}
        rows_mat = np.stack(rows_of_age) if uniform else None      # [A, E] original order (Kahan order)
        sorted_mat = np.stack(sorted_rows) if uniform else None    # [A, E] table order
        return place, page, cmin, cmax, contacts, rank, rows_of_age, sorted_rows, uniform, rows_mat, sorted_mat

    # main.pyx:1250-1266
    def set_mobility_factor(self, factor, place=None, min_age=None, max_age=None):
        factor = np.float32(factor)
        self.mobility_factor = factor
        if place is None:
            place = PLACE_ALL
        if min_age is None:
            min_age = 0
        if max_age is None:
            max_age = self.nr_ages - 1
        for mf in self.mobility_factors:
            if mf[0] == place and mf[1] == min_age and mf[2] == max_age:
                mf[3] = factor
                break
        else:
            self.mobility_factors.append([place, min_age, max_age, factor])
        self.mobility_factor_changed = True

    # main.pyx:1268-1283
    def set_mask_probability(self, p, place=None, min_age=None, max_age=None):
        if min_age is None:
            min_age = 0
        if max_age is None:
            max_age = self.nr_ages - 1
        places = list(range(len(PLACES))) if place is None else [place]
        lo, hi = max(min_age, 0), min(max_age, self.nr_ages - 1)
        if hi >= lo:
            for pl in places:
                self.mask_probabilities[lo:hi + 1, pl] = p

    # main.pyx:1285-1288
    def init_day(self):
        """Returns True when the tables were rebuilt (caller re-uploads them)."""
        if self.mobility_factor_changed:
            self.generate_contact_probabilities()
            self.mobility_factor_changed = False
            return True
        return False

    # main.pyx:1184-1235
    def _generate_native(self):
        """The uniform case through the library's host-side builder (reina_build_contact_tables, csrc/
        reina_contacts.h): the same arithmetic in C, ~20 us instead of ~0.8 ms of small numpy calls -- a
        day on which a mobility limitation changes must not leave the GPU waiting for its tables."""
        st = self._native_static
        if st is None:
            A, E = self._rows_mat.shape
            rf = self._sorted_mat.reshape(-1)
            st = self._native_static = dict(
                base=np.ascontiguousarray(self._contacts, dtype=np.float64),
                page=np.ascontiguousarray(self._page, dtype=np.int32),
                place=np.ascontiguousarray(self._place, dtype=np.int32),
                rows=np.ascontiguousarray(self._rows_mat, dtype=np.int32),
                sorted=np.ascontiguousarray(self._sorted_mat, dtype=np.int32),
                A=A, E=E, offset=(np.arange(A) * E).astype(np.int32), count=np.full(A, E, dtype=np.int32),
                places=self._place[rf].astype(np.int32), cmins=self._cmin[rf].astype(np.int32),
                cmaxs=self._cmax[rf].astype(np.int32), age_rep=np.repeat(np.arange(A), E))
        A, E = st['A'], st['E']
        mob = np.array([[-1.0 if m[0] == PLACE_ALL else float(m[0]), float(m[1]), float(m[2]), float(m[3])]
                        for m in self.mobility_factors], dtype=np.float64).reshape(-1, 4)
        totals = np.empty(A, dtype=np.float64)
        cum = np.empty(A * E, dtype=np.float64)
        nrc = np.zeros(self.pack_ages, dtype=np.float32)
        thr = np.empty((self.pack_ages, self.pack_entries), dtype=np.uint32)
        thr[A:] = 0xFFFFFFFF                 # the builder writes [0, A) x [0, E)
        if E < self.pack_entries:
            thr[:A, E:] = 0xFFFFFFFF
        rc = self.native_build(st['base'].ctypes.data, st['page'].ctypes.data, st['place'].ctypes.data, len(st['base']),
                               mob.ctypes.data, len(mob), st['rows'].ctypes.data, st['sorted'].ctypes.data, A, E,
                               totals.ctypes.data, cum.ctypes.data, nrc.ctypes.data, thr.ctypes.data, self.pack_entries)
        if rc != 0:
            raise RuntimeError('build_contact_tables failed (%d)' % rc)
        t = ContactTables(totals, st['offset'], st['count'], st['places'], st['cmins'], st['cmaxs'], cum, None)
        t.mask_source = (self.mask_probabilities.copy(), st['age_rep'], st['places'])   # mask_p on first use
        t.packed = (nrc, thr)   # pack_contact_tables takes these as they are
        return t

    def generate_contact_probabilities(self):
        if self._uniform and self.native_build is not None and self._rows_mat.shape[1] <= self.pack_entries:
            self.tables = self._generate_native()
            return self.tables
        contacts = self._contacts.copy()
        for place, min_age, max_age, factor in self.mobility_factors:
            if factor == 1.0:
                continue
            f = (self._page >= min_age) & (self._page <= max_age)
            if place != PLACE_ALL:
                f &= self._place == place
            contacts[f] *= float(factor)

        A = self.nr_ages
        if self._uniform:
            # all ages at once: Kahan-compensated sums (pandas groupby-sum) over the row axis
            c = contacts[self._rows_mat]                      # [A, E]
            sumx = np.zeros(A, dtype=np.float64)
            comp = np.zeros(A, dtype=np.float64)
            for k in range(c.shape[1]):
                y = c[:, k] - comp
                t = sumx + y
                comp = t - sumx - y
                sumx = t
            totals = sumx
            r = self._sorted_mat
            with np.errstate(divide='ignore', invalid='ignore'):
                cum = np.cumsum(contacts[r] / totals[:, None], axis=1)
            E = r.shape[1]
            offset = (np.arange(A) * E).astype(np.int32)
            count = np.full(A, E, dtype=np.int32)
            rf = r.reshape(-1)
            places = [self._place[rf]]
            cmins = [self._cmin[rf]]
            cmaxs = [self._cmax[rf]]
            cums = [cum.reshape(-1)]
            masks = [self.mask_probabilities[np.repeat(np.arange(A), E), self._place[rf]].astype(np.float32)]
        else:
            totals = np.zeros(A, dtype=np.float64)
            offset = np.zeros(A, dtype=np.int32)
            count = np.zeros(A, dtype=np.int32)
            places, cmins, cmaxs, cums, masks = [], [], [], [], []
            pos = 0
            for a in range(A):
                rows = self._rows_of_age[a]
                sumx = 0.0
                comp = 0.0
                for v in contacts[rows].tolist():  # pandas groupby(...).sum(): Kahan, row order
                    y = v - comp
                    t = sumx + y
                    comp = t - sumx - y
                    sumx = t
                totals[a] = sumx
                r = self._sorted_rows[a]
                with np.errstate(divide='ignore', invalid='ignore'):
                    cum = np.cumsum(contacts[r] / sumx)
                offset[a] = pos
                count[a] = len(r)
                pos += len(r)
                places.append(self._place[r])
                cmins.append(self._cmin[r])
                cmaxs.append(self._cmax[r])
                cums.append(cum)
                masks.append(self.mask_probabilities[a, self._place[r]].astype(np.float32))
        self.tables = ContactTables(
            totals, offset, count,
            np.concatenate(places).astype(np.int32), np.concatenate(cmins).astype(np.int32),
            np.concatenate(cmaxs).astype(np.int32), np.concatenate(cums).astype(np.float64),
            np.concatenate(masks).astype(np.float32))
        return self.tables
