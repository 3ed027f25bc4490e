"""Member-type selection by a genetic algorithm, with every generation's fitness evaluated in ONE
batched GPU solve (BASELINE config 4).

Same constructor, method names, fitness formula and - for a given `random.seed` - the same sequence
of `random` calls as the reference's `slientruss3d/ga.py:12-237`, so a seeded run walks the same
trajectory.  What differs: the reference's `Select` (`ga.py:155-160`) and `GetBestFeasibleGene`
(`ga.py:110-123`) call `Truss.Solve()` once per individual; here the population shares geometry,
supports and loads and differs only in (a, e, density) per member, so the whole population is one
`DeviceBatch` whose sections are gathered on the device from the member-type table and whose
weight / stress / displacement reductions run in `trs_fitness`.

`GetFitness(gene)` keeps working for one gene (and may be overridden, as the reference's docs allow,
`detail/truss_optimization.md:183-239`); an overridden `GetFitness` switches the population
evaluation back to one call per gene.
"""
import random

import numpy as np

from .truss import Truss
from .type import MemberType
from .utils import (INF, ZERO_EPS, EliteNumberTooMuchError, MinDisplaceTooLargeError,
                    MinStressTooLargeError, OnlyOneMemberTypeError, ProbabilityGreaterThanOneError)

PENALTY = 1e5  # weight of the normalised violations in the fitness (ga.py:146-148)


class GA:
    def __init__(self, truss: Truss, memberTypeList, allowStress=30000., allowDisplace=10.,
                 nIteration=None, nPatience=50, nPop=200, nElite=50, pCrossover=0.7, pMutate=0.1,
                 pOrigin=0.1, isCheckWorst=False, devices=None):
        """Arguments as the reference (`ga.py:14-29`).  `devices` (not in the reference): a list of GPU
        names - the population is then split over one worker process per GPU (128 per GPU for
        nPop = 1024 on 8 GPUs), geometry resident on every GPU, the fitness
        triples gathered on the host.  Default: the current device evaluates the whole population."""
        self.nPop, self.nElite = nPop, nElite
        self._devices = list(devices) if devices else None
        self._pool = None
        self.pCrossover, self.pMutate, self.pOrigin = pCrossover, pMutate, pOrigin
        self.pRandomGene = 1. - pCrossover - pMutate - pOrigin
        self.nIteration, self.nPatience = nIteration, nPatience
        self.truss = truss
        self.allowStress, self.allowDisplace = allowStress, allowDisplace
        self.typeList = memberTypeList
        self.nMember = truss.nMember
        self.nType = len(memberTypeList)
        self.memberIDList = truss.GetMemberIDs()
        self.memberIDMap = dict(enumerate(self.memberIDList))
        self._feasibleGene = [None] * self.nMember
        self._feasibleFitness = None
        self._device = None  # DeviceBatch of nPop copies, built on first use
        self.CheckRatioality(isCheckWorst)

    # ------------------------------------------------------------------ configuration checks
    @property
    def memberTypeWeightedInitProb(self):
        return [1.] * len(self.typeList)

    def CheckRatioality(self, isCheckWorst):
        """Parameter sanity (ga.py:70-108)."""
        if self.nElite > self.nPop:
            raise EliteNumberTooMuchError(
                f"Number of elites must <= number of population. Got [nElite] = {self.nElite}, [nPop] = {self.nPop}.")
        if self.pCrossover + self.pMutate + self.pOrigin > 1.:
            raise ProbabilityGreaterThanOneError(
                f"[pCrossover] + [pMutate] + [pOrigin] must <= 1.0, but got "
                f"[{self.pCrossover + self.pMutate + self.pOrigin :.4f}].")
        if self.nType <= 1:
            raise OnlyOneMemberTypeError(f"Number of member types must >= 2, but got {self.nType}.")
        if not isCheckWorst:
            return
        # the stiffest choices must satisfy the limits, otherwise no gene can (ga.py:86-108)
        byArea = max(self.typeList, key=lambda t: t.a)
        byStiffness = max(self.typeList, key=lambda t: t.e * t.a)
        self._setAll(byArea)
        self.truss.Solve()
        if not self.truss.IsInternalStressAllowed(self.allowStress)[0]:
            raise MinStressTooLargeError(
                "Minimum stress is too large. Need other member types which have more [A] value.")
        self._setAll(byStiffness)
        self.truss.Solve()
        if not self.truss.IsDisplacementAllowed(self.allowDisplace)[0]:
            raise MinDisplaceTooLargeError(
                "Minimum displacement is too large. Need other member types which have more [E*A] value.")

    def _setAll(self, memberType):
        for memberID in self.memberIDList:
            self.truss.SetMemberType(memberID, memberType)

    # ------------------------------------------------------------------------- gene helpers
    def TranslateGene(self, gene):
        return {self.memberIDMap[i]: self.typeList[locus] for i, locus in enumerate(gene)}

    def GetRandomGene(self):
        return random.choices(range(self.nType), k=self.nMember)

    def SetMemberTypesByGene(self, gene, truss):
        for i, locus in enumerate(gene):
            truss.SetMemberType(self.memberIDMap[i], self.typeList[locus])
        return truss

    def Initialize(self):
        weights = self.memberTypeWeightedInitProb
        return [random.choices(range(self.nType), k=self.nMember, weights=weights)
                for _ in range(self.nPop)]

    # ------------------------------------------------------------------------------ fitness
    def _compose(self, weight, stressViolation, displaceViolation):
        """Fitness triple from the three reductions (ga.py:143-149)."""
        okStress = abs(stressViolation) < ZERO_EPS
        okDisplace = abs(displaceViolation) < ZERO_EPS
        fitness = weight
        if not okStress:
            fitness += stressViolation / self.allowStress * PENALTY
        if not okDisplace:
            fitness += displaceViolation / self.allowDisplace * PENALTY
        return fitness, okStress, okDisplace

    def GetFitness(self, gene):
        """One gene (ga.py:139-149).  The population path does not call this unless overridden."""
        return self.GetFitnessBatch([gene])[0]

    def _population_device(self, count):
        from .batch import DeviceBatch, pack_trusses
        if self._device is None or self._device.B < count:
            import torch
            base = pack_trusses([self.truss])
            dev = self._device = DeviceBatch(base.replicate(max(count, self.nPop)))
            table = np.array([[t.a, t.e, t.density] for t in self.typeList], dtype=np.float64)
            self._typeTable = torch.from_numpy(table).to(dev.device)
            # a generation's traffic: the gene matrix up, three reductions + the status down - page-locked staging
            # on both sides, so that the call is two kernels between two asynchronous copies and ONE wait
            self._genesHost = torch.empty([dev.B, self.nMember], dtype=torch.uint8).pin_memory()
            self._genesDev = torch.empty([dev.B, self.nMember], dtype=torch.uint8, device=dev.device)
            self._fitDev = torch.empty([3, dev.B], dtype=torch.float64, device=dev.device)
            self._fitHost = torch.empty([3, dev.B], dtype=torch.float64).pin_memory()
            self._infoHost = torch.empty([dev.B], dtype=torch.int32).pin_memory()
        return self._device

    _POPULATION_STATE = ("_device", "_typeTable", "_genesHost", "_genesDev", "_fitDev", "_fitHost", "_infoHost")

    def _adopt_population(self, other):
        """Share `other`'s resident population batch and staging buffers (same truss, member types and nPop: a
        second GA over the same problem, e.g. another seed) instead of building them again."""
        for name in self._POPULATION_STATE:
            setattr(self, name, getattr(other, name))

    def _fitness_arrays(self, genes):
        """(fitness, isInternalAllowed, isDisplaceAllowed) as arrays for a population (list of genes or a gene
        matrix): gene matrix -> member sections (`trs_ga_sections`) -> solve + reductions (`trs_solve_small` with the
        GA terms on the fused path) on the resident population, one wait for the four result rows."""
        import torch
        if not self.truss.isStable:   # what Truss.Solve() raises per individual in the reference
            from .utils import TrussNotStableError
            raise TrussNotStableError("The truss is not stable !")
        count = len(genes)
        if self._devices is not None and len(self._devices) > 1:
            return self._fitness_sharded(genes)
        matrix = self._gene_matrix(genes)
        if matrix.size and (matrix.min() < 0 or matrix.max() >= self.nType):
            raise IndexError("a gene holds a locus outside the member type list")
        dev = self._population_device(count)
        with torch.cuda.device(dev.device):
            if self.nType <= 256:
                self._genesHost.numpy()[:count] = matrix
                self._genesDev.copy_(self._genesHost, non_blocking=True)
                dev.set_sections_from_genes(self._genesDev, count, self.nMember, self._typeTable)
            else:   # more types than a byte holds: gather with the tensor library
                loci = torch.zeros([dev.B, dev.nM_max], dtype=torch.int64, device=dev.device)
                loci[:count, :self.nMember] = torch.from_numpy(np.ascontiguousarray(matrix, dtype=np.int64)).to(dev.device)
                sections = self._typeTable[loci]                    # [B, nM, 3] = (a, e, density)
                dev.A.copy_(sections[..., 0]); dev.E.copy_(sections[..., 1]); dev.rho.copy_(sections[..., 2])
            dev.solve_fitness(self.allowStress, self.allowDisplace, out=list(self._fitDev.unbind(0)))
            self._fitHost.copy_(self._fitDev, non_blocking=True)
            self._infoHost.copy_(dev.info, non_blocking=True)
            torch.cuda.current_stream(dev.device).synchronize()
        if self._infoHost.numpy()[:count].any():
            raise np.linalg.LinAlgError("Singular matrix")
        terms = self._fitHost.numpy()
        return self._compose_arrays(terms[0, :count], terms[1, :count], terms[2, :count])

    def GetFitnessBatch(self, genes):
        """[(fitness, isInternalAllowed, isDisplaceAllowed)] for a list of genes: one batched solve.
        Member sections are gathered on the device from the type table by the gene matrix."""
        fitness, okStress, okDisplace = self._fitness_arrays(genes)
        return list(zip(fitness.tolist(), okStress.tolist(), okDisplace.tolist()))

    def _gene_matrix(self, genes):
        """The population (list of lists of type indices, the reference's representation) as an int64
        array [count, nMember].  `bytes()` of a gene is 8x faster than numpy's list-of-lists conversion,
        which was three quarters of the wall time of a generation.  A uint8 matrix (the native generation loop's
        own representation) passes through as it is."""
        if isinstance(genes, np.ndarray):
            return genes
        if self.nType <= 256:
            try:
                flat = np.frombuffer(b"".join(map(bytes, genes)), dtype=np.uint8)
                if flat.size == len(genes) * self.nMember:
                    return flat.reshape(len(genes), self.nMember).astype(np.int64)
            except (TypeError, ValueError):   # a gene that is not a list of small non-negative ints
                pass
        return np.asarray(genes, dtype=np.int64)

    def _compose_arrays(self, weight, stressViolation, displaceViolation):
        """`_compose` over arrays (same arithmetic, element by element): fitness, isInternalAllowed,
        isDisplaceAllowed."""
        okStress = np.abs(stressViolation) < ZERO_EPS
        okDisplace = np.abs(displaceViolation) < ZERO_EPS
        fitness = weight + np.where(okStress, 0.0, stressViolation / self.allowStress * PENALTY)
        fitness = fitness + np.where(okDisplace, 0.0, displaceViolation / self.allowDisplace * PENALTY)
        return fitness, okStress, okDisplace

    def _fitness_sharded(self, genes):
        """The population split over the GPUs of `devices` (`shard.ShardedSolver.fitness`)."""
        from .batch import pack_trusses
        from .shard import ShardedSolver
        if self._pool is None:
            self._pool = ShardedSolver(self._devices)
            self._base = pack_trusses([self.truss])
            self._table = np.array([[t.a, t.e, t.density] for t in self.typeList], dtype=np.float64)
        count = len(genes)
        pop = self._base.replicate(count)
        sec = self._table[self._gene_matrix(genes)]                    # [count, nMember, 3]
        pop.A[:, :self.nMember], pop.E[:, :self.nMember] = sec[..., 0], sec[..., 1]
        pop.rho[:, :self.nMember] = sec[..., 2]
        fit, info = self._pool.fitness(pop, self.allowStress, self.allowDisplace, geometry_key=id(self))
        if info.any():
            raise np.linalg.LinAlgError("Singular matrix")
        return self._compose_arrays(fit[:, 0], fit[:, 1], fit[:, 2])

    def close(self):
        """Stop the per-GPU worker processes of a multi-device GA (no-op otherwise)."""
        if self._pool is not None:
            self._pool.close()
            self._pool = None

    def _evaluate(self, pop):
        """Population -> list of fitness triples; batched unless GetFitness was overridden."""
        if type(self).GetFitness is not GA.GetFitness:
            return [self.GetFitness(gene) for gene in pop]
        return self.GetFitnessBatch(pop)

    # ---------------------------------------------------------------------------- operators
    def _RecordFeasible(self, evaluatedPop, isSorted=False):
        for gene, (fitness, okStress, okDisplace) in evaluatedPop:
            if okStress and okDisplace and (self._feasibleFitness is None or fitness < self._feasibleFitness):
                self._feasibleGene[:], self._feasibleFitness = gene, fitness
                if isSorted:
                    break

    def Select(self, pop, isRecordFeasible=False):
        """Rank the population (stable sort by fitness, ga.py:155-160) and return the elites."""
        ranked = sorted(([gene, info] for gene, info in zip(pop, self._evaluate(pop))),
                        key=lambda pair: pair[1][0])
        if isRecordFeasible:
            self._RecordFeasible(ranked, isSorted=True)
        return [gene for gene, _ in ranked[:self.nElite]], ranked[0][1]

    def Crossover(self, gene0, gene1):
        cut0, cut1 = sorted(random.sample(range(self.nMember), k=2))
        return [gene0[i] if i < cut0 or i >= cut1 else gene1[i] for i in range(self.nMember)]

    def Mutate(self, gene):
        child = gene.copy()
        at = random.randint(0, self.nMember - 1)
        child[at] = random.choice([t for t in range(self.nType) if t != child[at]])
        return child

    def UpdatePop(self, pop, elitePop):
        """Next generation (ga.py:173-190): elites, then one random draw per remaining slot."""
        toCross = self.pCrossover
        toMutate = toCross + self.pMutate
        toKeep = toMutate + self.pOrigin
        newPop = list(elitePop) + [None] * (self.nPop - self.nElite)
        for j in range(self.nElite, self.nPop):
            p = random.random()
            if p <= toCross:
                newPop[j] = self.Crossover(*random.sample(elitePop, k=2))
            elif p <= toMutate:
                newPop[j] = self.Mutate(random.choice(elitePop))
            elif p <= toKeep:
                newPop[j] = pop[j]
            else:
                newPop[j] = self.GetRandomGene()
        return newPop

    def GetBestFeasibleGene(self, pop, isDirectlyReturnRecord=False):
        if isDirectlyReturnRecord and self._feasibleFitness is not None:
            return self._feasibleGene, (self._feasibleFitness, True, True)
        best, bestInfo = None, (INF, False, False)
        for gene, info in zip(pop, self._evaluate(pop)):
            if info[1] and info[2] and info[0] < bestInfo[0]:
                best, bestInfo = gene, info
        if best is None and self._feasibleFitness is not None:
            return self._feasibleGene, (self._feasibleFitness, True, True)
        return best, bestInfo

    # ------------------------------------------------------------- native generation loop
    #: the methods whose reference behaviour `_EvolveNative` reproduces; a subclass that overrides any of them
    #: (the reference's documented extension point is GetFitness) gets the plain Python loop
    _NATIVE_METHODS = ("GetFitness", "Select", "Crossover", "Mutate", "UpdatePop", "GetRandomGene",
                       "GetBestFeasibleGene", "_RecordFeasible", "_evaluate", "_compose_arrays", "_fitness_arrays")

    def _native_loop_ok(self):
        """The whole generation loop can run on gene MATRICES with the native population update (`csrc/gaops.c`):
        stock methods, at most 256 member types, CPython's Mersenne Twister behind `random`, the host library."""
        if any(getattr(type(self), m) is not getattr(GA, m) for m in self._NATIVE_METHODS):
            return False
        if not (2 <= self.nType <= 256 and self.nElite >= 2 and self.nMember >= 2):
            return False
        if self._devices is not None and len(self._devices) > 1:
            return False
        state = random.getstate()
        if not (state[0] == 3 and len(state[1]) == 625 and type(random._inst) is random.Random):
            return False
        try:
            from .generate import _load
            return hasattr(_load(), "trs_ga_update_pop")
        except Exception:
            return False

    def _update_pop_native(self, pop, elite, out):
        """`UpdatePop` on uint8 gene matrices (`trs_ga_update_pop`): the draws come out of Python's global generator
        - state handed over and put back -, in the reference's order (ga.py:173-190)."""
        import ctypes
        from .generate import _load
        version, words, gauss = random.getstate()
        state = np.array(words, dtype=np.uint32)
        toCross = self.pCrossover
        toMutate = toCross + self.pMutate
        toKeep = toMutate + self.pOrigin
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        lib = _load()
        lib.trs_ga_update_pop.restype = ctypes.c_int
        rc = lib.trs_ga_update_pop(ptr(state), ctypes.c_int(self.nPop), ctypes.c_int(self.nElite),
                                   ctypes.c_int(self.nMember), ctypes.c_int(self.nType), ctypes.c_double(toCross),
                                   ctypes.c_double(toMutate), ctypes.c_double(toKeep), ptr(elite), ptr(pop), ptr(out),
                                   None)
        if rc != 0:
            raise RuntimeError(f"trs_ga_update_pop refused the population ({rc})")
        random.setstate((version, tuple(state.tolist()), gauss))

    def _EvolveNative(self, isPrintMessage):
        """`Evolve` with the population as ONE uint8 matrix [nPop, nMember]: per generation one batched fitness
        call on the GPU, a stable argsort (= the reference's stable `sorted`, ga.py:157), the feasible-record
        update on arrays and the native `UpdatePop`.  Same `random` consumption as the Python loop, so the same
        trajectory; returns the reference's list-of-lists population."""
        pop = np.array(self.Initialize(), dtype=np.uint8)
        nxt = np.empty_like(pop)
        bestFitness, history, nWait, earlyStop = INF, [], 0, False
        iteration = 0

        stock_batch = type(self).GetFitnessBatch is GA.GetFitnessBatch

        def evaluate(matrix):
            if stock_batch:
                return self._fitness_arrays(matrix)
            # (a user's own batch evaluator gets the reference's list-of-lists population)
            info = self.GetFitnessBatch(matrix.tolist())   # [(fitness, okStress, okDisplace)]
            return (np.fromiter((t[0] for t in info), dtype=np.float64, count=len(info)),
                    np.fromiter((t[1] for t in info), dtype=bool, count=len(info)),
                    np.fromiter((t[2] for t in info), dtype=bool, count=len(info)))

        while self.nIteration is None or iteration < self.nIteration:
            fit, okS, okD = evaluate(pop)
            order = np.argsort(fit, kind="stable")
            feasible = np.flatnonzero((okS & okD)[order])
            if feasible.size:                         # the best feasible gene of this generation (ranked order)
                b = int(order[feasible[0]])
                if self._feasibleFitness is None or float(fit[b]) < self._feasibleFitness:
                    self._feasibleGene[:], self._feasibleFitness = pop[b].tolist(), float(fit[b])
            first = int(order[0])
            minFitness, okStress, okDisplace = float(fit[first]), bool(okS[first]), bool(okD[first])
            if minFitness < bestFitness:
                bestFitness, nWait = minFitness, 0
            else:
                nWait += 1
                if nWait >= self.nPatience:
                    earlyStop = True
                    break
            history.append(bestFitness)
            if isPrintMessage:
                print(f"\rIteration: {iteration :6d}, nWaitBestIter: {nWait :3d}, minFitness: {minFitness :12.4f}, "
                      f"isInternalAllowed: {str(okStress) :5s}, isDisplaceAllowed: {str(okDisplace) :5s}", end='')
            elite = np.ascontiguousarray(pop[order[:self.nElite]])
            self._update_pop_native(pop, elite, nxt)
            pop, nxt = nxt, pop
            iteration += 1
        if isPrintMessage:
            print('...Early stoping !' if earlyStop else "")
        popList = pop.tolist()
        minGene, minGeneInfo = self.GetBestFeasibleGene(popList, earlyStop)
        if minGene is None:
            minGene = popList[0]
            minGeneInfo = self._evaluate([minGene])[0]
            if isPrintMessage:
                print('-' * 50 + '\n' + "Warning: Cannot find any feasible result, so only return the gene "
                      "which has lowest fitness." + '\n' + '-' * 50)
        return minGene, minGeneInfo, popList, history

    # -------------------------------------------------------------------------------- driver
    def Evolve(self, isPrintMessage=True, native=None):
        """Run the GA (ga.py:192-237).  Returns (minGene, minGeneInfo, finalPop, bestFitnessHistory).
        `native` (not in the reference): None = use the native generation loop (`_EvolveNative`: gene matrix +
        `csrc/gaops.c`) whenever no method it replaces is overridden, False = the plain Python loop; both draw the
        same numbers from `random` in the same order."""
        if native is None:
            native = self._native_loop_ok()
        elif native and not self._native_loop_ok():
            raise ValueError("the native generation loop needs the stock GA methods, <= 256 member types and the host library")
        if native:
            return self._EvolveNative(isPrintMessage)
        pop = self.Initialize()
        bestFitness, history, nWait, earlyStop = INF, [], 0, False
        iteration = 0
        while self.nIteration is None or iteration < self.nIteration:
            elitePop, (minFitness, okStress, okDisplace) = self.Select(pop, True)
            if minFitness < bestFitness:
                bestFitness, nWait = minFitness, 0
            else:
                nWait += 1
                if nWait >= self.nPatience:
                    earlyStop = True
                    break
            history.append(bestFitness)
            if isPrintMessage:
                print(f"\rIteration: {iteration :6d}, nWaitBestIter: {nWait :3d}, minFitness: {minFitness :12.4f}, "
                      f"isInternalAllowed: {str(okStress) :5s}, isDisplaceAllowed: {str(okDisplace) :5s}", end='')
            pop = self.UpdatePop(pop, elitePop)
            iteration += 1
        if isPrintMessage:
            print('...Early stoping !' if earlyStop else "")
        minGene, minGeneInfo = self.GetBestFeasibleGene(pop, earlyStop)
        if minGene is None:
            minGene = pop[0]
            minGeneInfo = self._evaluate([minGene])[0]
            if isPrintMessage:
                print('-' * 50 + '\n' + "Warning: Cannot find any feasible result, so only return the gene "
                      "which has lowest fitness." + '\n' + '-' * 50)
        return minGene, minGeneInfo, pop, history
