"""The slice of mfrec/recommendation/base.py the training path needs: the ratings
container, parameter handling, ratings -> kernel-input marshalling, model snapshots and the
save_state / load_state file set (so that factors trained here load into the original tooling
and vice versa).  Database backends, similarity search and KNN predictors are out of scope
(SURVEY 8)."""
import logging
import pickle

import numpy as np
import scipy.sparse as sp
from scipy.sparse import lil_matrix


class Error(Exception):
    """Raised for unknown parameter names (base.py:23, :199)."""


class BaseRecommender(object):
    PARAMETERS_INDEX = {}
    _logger_name = "mfrec.recommender"

    def __init__(self, nbr_users=4, nbr_items=6, parameters=None):
        self.logger = logging.getLogger(self._logger_name)
        self.dimensionality = 40                      # base.py:90
        self.min_rating, self.max_rating = 1.0, 5.0   # base.py:92-93
        self._nbr_users, self._nbr_items = int(nbr_users), int(nbr_items)
        self.relationship_matrix = None
        self._csc, self._csc_due = None, False
        self.relationship_matrix_csr = None
        self.items_feedback = self.feedback_rated = self.feedback_hash = None
        self.svd_u = self.svd_v = self.svd_s = None
        self.users_bias = self.items_bias = self.overall_bias = None
        # label <-> index maps of the reference's container (base.py:96-101): label LISTS indexed by id,
        # index DICTS keyed by label; carried through save_state
        self.users_label, self.users_index = [], {}
        self.items_label, self.items_index, self.items_index_map = [], {}, {}
        self.initialize_relationship_matrix(self._nbr_users, self._nbr_items)

    def initialize_relationship_matrix(self, nbr_users, nbr_items):
        """base.py:266-281: an empty users x items lil_matrix and the default labels 'user<i>' / 'item<i>'."""
        self.relationship_matrix = lil_matrix((nbr_users, nbr_items))
        for item in range(nbr_items):
            self.items_label.append("item" + str(item))
            self.items_index["item" + str(int(item))] = int(item)
        for user in range(nbr_users):
            self.users_label.append("user" + str(user))
            self.users_index["user" + str(user)] = user

    @property
    def relationship_matrix_csc(self):
        """items x users CSC view of the ratings (the trainers set `relationship_matrix.T.tocsc()`, kmf.py:207,
        wrmf.py:88).  Formed on first use after a train() -- at 10^7 ratings the conversion costs more than the
        training it precedes."""
        if self._csc is None and self._csc_due:
            self._csc = self.relationship_matrix.T.tocsc()
            self._csc_due = False
        return self._csc

    @relationship_matrix_csc.setter
    def relationship_matrix_csc(self, value):
        self._csc, self._csc_due = value, False

    def _mark_csc_due(self):
        self._csc, self._csc_due = None, True

    @property
    def nbr_users(self):
        return self._nbr_users

    @property
    def nbr_items(self):
        return self._nbr_items

    def set_parameters(self, parameters):
        """public name -> attribute through PARAMETERS_INDEX (base.py:180-199)."""
        for name, value in parameters.items():
            if name not in self.PARAMETERS_INDEX:
                raise Error("Wrong parameters")
            setattr(self, self.PARAMETERS_INDEX[name], value)

    # ---- ratings container ----------------------------------------------------------
    def set_item_by_id(self, user_index, item_index, value):
        self.relationship_matrix[user_index, item_index] = float(value)     # base.py:823-828

    def set_item(self, user, items_list):
        """base.py:815-820: items_list = [{'label': item label, 'value': rating}, ...] for the user LABEL."""
        for item in items_list:
            self.relationship_matrix[int(self.users_index[user]), int(self.items_index[item["label"]])] = float(item["value"])

    def set_item_by_label(self, user, item, value):                          # base.py:831-836
        self.relationship_matrix[int(self.users_index[user]), int(self.items_index[item])] = float(value)

    def build_index(self):
        """base.py:839-851: rebuild both index dicts from the label lists."""
        self.users_index = {}
        self.items_index = {}
        for i, user in enumerate(self.users_label):
            self.users_index[user] = i
        for i, item in enumerate(self.items_label):
            self.items_index[item] = i

    def set_user_label(self, user_index, label):                             # base.py:1097-1103
        del self.users_index[self.users_label[user_index]]
        self.users_index[label] = user_index
        self.users_label[user_index] = label

    def set_item_label(self, item_index, label):                             # base.py:1106-1112
        del self.items_index[self.items_label[item_index]]
        self.items_index[label] = item_index
        self.items_label[item_index] = label

    def _get_new_user_id(self):                                              # base.py:1053-1060
        new_id = len(self.users_label)
        self.users_label.append("user" + str(new_id))
        return new_id

    def _get_new_item_id(self):
        """base.py:1043-1050 takes len(users_label) here -- a slip in a method marked @todo; the item list's
        own length is what keeps label list and factor columns in step."""
        new_id = len(self.items_label)
        self.items_label.append("item" + str(new_id))
        return new_id

    def set_dimensionality(self, new_dim_value):                             # base.py:289-290
        self.dimensionality = new_dim_value

    def set_ratings(self, ratings_index, ratings):
        """Bulk load of (user, item) -> rating triples: the matrix a loop of set_item_by_id calls over the triples
        would leave (later duplicates win, a cell that is given again is overwritten, a 0 clears its cell),
        built with array operations -- the per-cell loop needs ~2.5 us per rating (base.py:823-828 is one lil
        assignment per call), i.e. minutes at 10^8 ratings."""
        ratings_index = np.asarray(ratings_index)
        vals = np.asarray(ratings, dtype=np.float64)
        if ratings_index.ndim != 2 or ratings_index.shape[1] != 2 or ratings_index.shape[0] != vals.shape[0]:
            raise Error("The index and the ratings array must be the same size")
        n = vals.shape[0]
        if n == 0:
            return
        u = ratings_index[:, 0].astype(np.int64)
        i = ratings_index[:, 1].astype(np.int64)
        if u.min() < 0 or u.max() >= self.nbr_users or i.min() < 0 or i.max() >= self.nbr_items:
            raise IndexError("rating outside the %d x %d matrix" % (self.nbr_users, self.nbr_items))
        keys = u * self.nbr_items + i
        _, first_rev = np.unique(keys[::-1], return_index=True)         # last occurrence of every cell
        last = n - 1 - first_rev
        shape = (self.nbr_users, self.nbr_items)
        new = sp.csr_matrix((vals[last], (u[last], i[last])), shape=shape)
        old = self.relationship_matrix.tocsr()
        if old.nnz:
            given = sp.csr_matrix((np.ones(last.shape[0]), (u[last], i[last])), shape=shape)
            old = old - old.multiply(given)                             # cells given again lose their old value
            new = old + new
        new.eliminate_zeros()
        self.relationship_matrix = new.tolil()

    def get_nbr_ratings(self):
        return int(self.relationship_matrix.tocoo().nnz)

    def ratings_iterator(self):
        coo = self.relationship_matrix.tocoo()                               # base.py:284-286
        return zip(coo.row, coo.col, coo.data)

    def get_ratings(self, randomize_order=False):
        """-> (int32 [nnz,2] (user,item), f64 [nnz]) in COO order of the lil matrix (users
        ascending, items ascending); optionally shuffled with the global numpy RNG exactly
        where the reference shuffles (base.py:1115-1131)."""
        coo = self.relationship_matrix.tocoo()
        keep = coo.data != 0
        ratings_index = np.stack([coo.row[keep], coo.col[keep]], axis=1).astype(np.int32)
        ratings = coo.data[keep].astype(np.float64)
        order = np.arange(ratings.shape[0])
        if isinstance(randomize_order, str) and randomize_order == "balanced":
            # extension: a random order with the fewest dependency levels (seeded from the same
            # global numpy RNG the reference's shuffle uses); falls back to the plain shuffle
            from ..lib.datasets import balanced_order, balanced_order_gpu
            from .._shim import MfxError
            # what a builder's refusal looks like after _shim.check(): MFX_E_STATE -> MfxError (too many
            # levels), MFX_E_NOMEM -> MemoryError (colour bitsets / scratch), never a wrong order
            refusal = (MfxError, MemoryError)
            np.random.shuffle(order)               # the random part, from the RNG the reference uses
            try:
                try:                               # GPU builder: milliseconds (recolours the shuffled order)
                    perm, _ = balanced_order_gpu(ratings_index[order], self.nbr_users, self.nbr_items)
                    order = order[perm]
                except refusal:                    # > 2^20 levels, bitsets too large: the host builder
                    order, _ = balanced_order(ratings_index, self.nbr_users, self.nbr_items,
                                              seed=int(np.random.randint(0, 2 ** 31 - 1)))
            except refusal:
                pass                               # keep the plain shuffle
        elif randomize_order:
            np.random.shuffle(order)
        return np.ascontiguousarray(ratings_index[order]), np.ascontiguousarray(ratings[order])

    def get_feedback(self):
        """base.py:269-296: the rated (user, item) pairs in COO order of the boolean ratings matrix
        (users ascending, items ascending) and, per user, (first row, count) -- (-1, -1) for a user without ratings."""
        coo = self.relationship_matrix.tocoo()
        keep = coo.data != 0
        pairs = np.stack([coo.row[keep], coo.col[keep]], axis=1).astype(np.int32)
        pairs = pairs[np.lexsort((pairs[:, 1], pairs[:, 0]))]
        counts = np.bincount(pairs[:, 0], minlength=self.nbr_users).astype(np.int64)
        first = np.concatenate([[0], np.cumsum(counts)[:-1]])
        rated_hash = np.stack([np.where(counts > 0, first, -1), np.where(counts > 0, counts, -1)], axis=1).astype(np.int32)
        return np.ascontiguousarray(pairs), np.ascontiguousarray(rated_hash)

    def initialize_rated_feedback(self):                                      # base.py:261-263
        self.N = self.relationship_matrix.astype(bool)
        self.feedback_rated, self.feedback_hash = self.get_feedback()

    def get_items_subset(self, count=100, method="random"):
        """base.py:1134-1145: `count` item ids of a shuffle drawn from the global numpy RNG."""
        ids = np.arange(self.nbr_items)
        np.random.shuffle(ids)
        return ids[0:count]

    def compute_overall_avg(self):
        data = self.relationship_matrix.tocoo().data                         # base.py:504-508
        self.overall_bias = float(data[data != 0].astype(float).mean())

    # ---- save_state / load_state: the file set of base.py:557-596 ----------------------
    _STATE_DICTS = (("_userslabel.dat", "users_label"), ("_usersindex.dat", "users_index"),
                    ("_itemslabel.dat", "items_label"), ("_itemsindex.dat", "items_index"),
                    ("_itemsindexmap.dat", "items_index_map"))

    def save_state(self, filename):
        """`filename`_relations.mtx (MatrixMarket), `filename`_svd.npz with keys svd_s / svd_u / svd_v
        ([k, n] f64; svd_s is None for the SGD / ALS families and is stored the way np.savez stores None,
        exactly what the reference writes), and the five label / index dictionaries as pickles --
        protocol 0, which the reference's Python 2 reads (base.py:557-575)."""
        import scipy.io
        scipy.io.mmwrite(filename + "_relations.mtx", self.relationship_matrix)
        np.savez(filename + "_svd.npz", svd_s=self.svd_s, svd_u=self.svd_u, svd_v=self.svd_v)
        for suffix, attr in self._STATE_DICTS:
            with open(filename + suffix, "wb") as f:
                pickle.dump(getattr(self, attr), f, protocol=0)
        self.logger.info("The state of the recommendation engine has been saved.")

    def load_state(self, filename):
        """Inverse of save_state; also reads a file set written by the reference itself (its pickles are
        Python 2 text pickles, its svd_s an object array): base.py:578-596.  The ratings come back as a
        lil_matrix (the reference keeps mmread's coo_matrix; every consumer here indexes the matrix).
        TRUST: the five dictionaries are pickles (the reference's format): loading a state executes whatever the files' author
        put there -- only load files you wrote yourself.  The members of `_svd.npz` are read WITHOUT a general unpickle: a trained
        factor is a plain f64 array; a member stored as an object array may only be np.savez's image of None (svd_s of the SGD /
        ALS families; svd_u / svd_v of a recommender saved before training) and is decoded by an unpickler that admits numpy's
        three array constructors and refuses every other global -- anything else in it is an error, not executed."""
        import io
        import zipfile
        import scipy.io
        z = np.load(filename + "_svd.npz", allow_pickle=False)

        class _NoGlobals(pickle.Unpickler):
            # np.save pickles the 0-d object ARRAY (three numpy constructors), its element inline: None needs no global at all
            _ARRAY = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                      ("numpy", "ndarray"), ("numpy", "dtype")}

            def find_class(self, module, name):
                if (module, name) in self._ARRAY:
                    return super().find_class(module, name)
                raise pickle.UnpicklingError("%s_svd.npz: object member names %s.%s -- refused" % (filename, module, name))

        def none_only(key):
            with zipfile.ZipFile(filename + "_svd.npz") as zf, zf.open(key + ".npy") as f:
                version = np.lib.format.read_magic(f)
                shape, _, dtype = (np.lib.format.read_array_header_1_0(f) if version == (1, 0)
                                   else np.lib.format.read_array_header_2_0(f))
                if dtype != object or shape != ():
                    raise ValueError("%s_svd.npz: member %s is an object array of shape %r; only np.savez's image of None is "
                                     "accepted" % (filename, key, shape))
                value = _NoGlobals(io.BytesIO(f.read()), encoding="latin1").load()
            if isinstance(value, np.ndarray) and value.dtype == object and value.shape == ():
                value = value.item()
            if value is not None:
                raise ValueError("%s_svd.npz: member %s holds a pickled %s, not None -- refused" % (filename, key, type(value).__name__))
            return None

        def member(key):
            try:
                return z[key]                        # a plain array: no pickle involved
            except ValueError:                       # "Object arrays cannot be loaded when allow_pickle=False"
                return none_only(key)

        self.svd_u, self.svd_v, self.svd_s = member("svd_u"), member("svd_v"), member("svd_s")
        self.relationship_matrix = lil_matrix(scipy.io.mmread(filename + "_relations.mtx"))
        self._nbr_users, self._nbr_items = self.relationship_matrix.shape
        for suffix, attr in self._STATE_DICTS:
            with open(filename + suffix, "rb") as f:
                setattr(self, attr, pickle.load(f, encoding="latin1"))
        self.logger.info("The state of the recommendation engine has been loaded")

    # ---- model snapshots (npz keys as base.py:805-812) -------------------------------
    def save_model_snapshot(self, filename):
        np.savez(filename + "_model_snapshot.npz", svd_u=self.svd_u, svd_v=self.svd_v)

    def load_model_snapshot(self, filename):
        z = np.load(filename + "_model_snapshot.npz")
        self.svd_u, self.svd_v = z["svd_u"], z["svd_v"]
