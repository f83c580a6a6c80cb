//! Sharded proving: ONE joint proof over the GPUs of a node, one AIR (trace) per GPU, one process
//! per GPU (DESIGN.md §6 of the HIP backend). Each rank owns the traces it generated
//! and the cached-main commitments of those traces; the per-round sumcheck values travel through the
//! library's shared-memory exchange, and the one bulk exchange -- the sum of the ranks' mu-batched WHIR
//! vectors -- goes through the `allgather_device` / `alltoall_device` callbacks, which a deployment
//! implements with RCCL (`ncclAllGather` / `ncclAllToAll` on the rank's stream). The reference has no
//! peer for this file.
use std::{
    ffi::{c_void, CString},
    os::raw::c_int,
    ptr,
};

use openvm_stark_backend::prover::{DeviceMultiStarkProvingKey, ProvingContext};

use crate::{
    device::abi_traces,
    error::HipError,
    ffi,
    transcript::DuplexSpongeHip,
    HipBackend, HipDevice,
};

/// The device all-gather a deployment supplies (RCCL over xGMI): `d_all[r * bytes ..] = rank r's d_mine`,
/// ordered after the work already enqueued on `stream` and complete for later work on it on return.
pub trait BulkExchange: Send {
    fn allgather_device(&mut self, d_mine: *const c_void, d_all: *mut c_void, bytes: u64, stream: ffi::OvmStream) -> i32;
    /// `d_recv[s * bytes ..] = rank s's d_send[rank * bytes ..]`; `None` when the deployment has no
    /// all-to-all (the prover then all-gathers whole vectors: N-1 instead of 2 (N-1)/N vectors per GPU).
    fn alltoall_device(
        &mut self,
        _d_send: *const c_void,
        _d_recv: *mut c_void,
        _bytes: u64,
        _stream: ffi::OvmStream,
    ) -> Option<i32> {
        None
    }
    /// Whether `alltoall_device` is implemented (decides if the callback is handed to the library).
    fn has_alltoall(&self) -> bool {
        false
    }
}

/// One rank's communicator: the shared-memory segment for the small exchanges + the bulk callback.
pub struct ShardComm {
    shm: *mut ffi::OvmShmComm,
    bulk: Box<dyn BulkExchange>,
    pub rank: u32,
    pub world: u32,
}

unsafe extern "C" fn host_trampoline(ctx: *mut c_void, mine: *const c_void, all: *mut c_void, bytes: u64) -> c_int {
    let comm = &*(ctx as *const ShardComm);
    ffi::ovm_hip_shm_allgather(comm.shm as *mut c_void, mine, all, bytes)
}
/// `ovm_shard_t.abort`: this rank failed in the middle of a joint call; the peers stop waiting and every rank's
/// communicator is poisoned (see `ShardComm::poisoned`).
unsafe extern "C" fn abort_trampoline(ctx: *mut c_void, code: c_int) -> c_int {
    let comm = &*(ctx as *const ShardComm);
    ffi::ovm_hip_shm_comm_abort(comm.shm as *mut c_void, code)
}
unsafe extern "C" fn device_trampoline(
    ctx: *mut c_void,
    d_mine: *const c_void,
    d_all: *mut c_void,
    bytes: u64,
    stream: ffi::OvmStream,
) -> c_int {
    let comm = &mut *(ctx as *mut ShardComm);
    comm.bulk.allgather_device(d_mine, d_all, bytes, stream)
}

unsafe extern "C" fn alltoall_trampoline(
    ctx: *mut c_void,
    d_send: *const c_void,
    d_recv: *mut c_void,
    bytes: u64,
    stream: ffi::OvmStream,
) -> c_int {
    let comm = &mut *(ctx as *mut ShardComm);
    comm.bulk.alltoall_device(d_send, d_recv, bytes, stream).unwrap_or(1)
}

/// The library's own bulk exchange (`csrc/ipc_comm.cpp`): peers map each other's send buffers through
/// HIP IPC and pull their slices with device-to-device copies -- over xGMI between the GPUs of a node. No
/// collective library is linked; `ShardComm::with_ipc` builds a communicator around it.
pub struct IpcBulk {
    ipc: *mut ffi::OvmIpcComm,
}
// SAFETY: the communicator is only used from the thread that drives the prover.
unsafe impl Send for IpcBulk {}
impl BulkExchange for IpcBulk {
    fn allgather_device(&mut self, d_mine: *const c_void, d_all: *mut c_void, bytes: u64, stream: ffi::OvmStream) -> i32 {
        // SAFETY: device pointers of this process, valid for `bytes` / `world * bytes`.
        unsafe { ffi::ovm_hip_ipc_allgather_device(self.ipc as *mut c_void, d_mine, d_all, bytes, stream) }
    }
    fn alltoall_device(&mut self, d_send: *const c_void, d_recv: *mut c_void, bytes: u64, stream: ffi::OvmStream) -> Option<i32> {
        // SAFETY: as above.
        Some(unsafe { ffi::ovm_hip_ipc_alltoall_device(self.ipc as *mut c_void, d_send, d_recv, bytes, stream) })
    }
    fn has_alltoall(&self) -> bool {
        true
    }
}
/// Which carrier the library's IPC exchange is on (`ovm_hip_ipc_comm_status`): mapped peer buffers, or -- after a failed or
/// overdue `hipIpcOpenMemHandle`, agreed by all ranks -- staged through host memory, and why.
#[derive(Debug, Clone)]
pub struct IpcStatus {
    pub degraded: bool,
    pub mappings: u32,
    pub opens: u64,
    pub exchanges: u64,
    pub staged_exchanges: u64,
    /// exchanges whose send buffer lay in an allocation the runtime cannot map (bit 31 of its size) and went out as a copy
    pub bounced_exchanges: u64,
    pub reason: String,
}

impl IpcBulk {
    pub fn status(&self) -> Result<IpcStatus, HipError> {
        // SAFETY: `ipc` is a live communicator; the library fills the POD it is handed.
        let mut st: ffi::OvmIpcStatus = unsafe { std::mem::zeroed() };
        HipError::from_result("ovm_hip_ipc_comm_status", unsafe { ffi::ovm_hip_ipc_comm_status(self.ipc, &mut st) })?;
        let reason = unsafe { std::ffi::CStr::from_ptr(st.reason.as_ptr()) }.to_string_lossy().into_owned();
        Ok(IpcStatus {
            degraded: st.degraded != 0,
            mappings: st.mappings,
            opens: st.opens,
            exchanges: st.exchanges,
            staged_exchanges: st.staged_exchanges,
            bounced_exchanges: st.bounced_exchanges,
            reason,
        })
    }
}

impl Drop for IpcBulk {
    fn drop(&mut self) {
        // SAFETY: created by ovm_hip_ipc_comm_create, destroyed once -- before the shared-memory communicator it
        // uses (ShardComm::drop replaces `bulk` first).
        unsafe { ffi::ovm_hip_ipc_comm_destroy(self.ipc) };
    }
}

impl ShardComm {
    /// A communicator whose bulk exchange is the library's own (no RCCL binding needed).
    pub fn with_ipc(name: &str, rank: u32, world: u32) -> Result<Box<Self>, HipError> {
        struct Pending;
        impl BulkExchange for Pending {
            fn allgather_device(&mut self, _: *const c_void, _: *mut c_void, _: u64, _: ffi::OvmStream) -> i32 {
                1
            }
        }
        let mut comm = Self::new(name, rank, world, Box::new(Pending))?;
        let mut ipc: *mut ffi::OvmIpcComm = ptr::null_mut();
        // SAFETY: `comm.shm` stays valid for the life of `comm`, which owns the IpcBulk.
        HipError::from_result("ovm_hip_ipc_comm_create", unsafe { ffi::ovm_hip_ipc_comm_create(comm.shm, rank, world, &mut ipc) })?;
        comm.bulk = Box::new(IpcBulk { ipc });
        Ok(comm)
    }

    /// `name`: the same unique shared-memory name on every rank of the job (e.g. "/ovm-<port>-<pid>").
    pub fn new(name: &str, rank: u32, world: u32, bulk: Box<dyn BulkExchange>) -> Result<Box<Self>, HipError> {
        let cname = CString::new(name).expect("shared-memory name with a NUL byte");
        let mut shm: *mut ffi::OvmShmComm = ptr::null_mut();
        // SAFETY: valid out-pointer; returns when every rank has attached.
        HipError::from_result("ovm_hip_shm_comm_create", unsafe {
            ffi::ovm_hip_shm_comm_create(cname.as_ptr(), rank, world, 1 << 20, &mut shm)
        })?;
        Ok(Box::new(Self { shm, bulk, rank, world }))
    }
    fn abi(self: &mut Box<Self>) -> ffi::OvmShard {
        ffi::OvmShard {
            rank: self.rank,
            world: self.world,
            ctx: (&mut **self) as *mut ShardComm as *mut c_void,
            allgather_host: Some(host_trampoline),
            allgather_device: Some(device_trampoline),
            alltoall_device: if self.bulk.has_alltoall() { Some(alltoall_trampoline) } else { None },
            abort: Some(abort_trampoline),
        }
    }
    /// A joint call over this communicator timed out or was aborted by a peer: the ranks' sequence numbers no longer
    /// agree; drop it and create a new one (every later call fails with `OVM_ESHARD_POISONED`).
    pub fn poisoned(&self) -> bool {
        // SAFETY: `shm` is valid for the life of `self`.
        unsafe { ffi::ovm_hip_shm_comm_poisoned(self.shm) != 0 }
    }
}
impl Drop for ShardComm {
    fn drop(&mut self) {
        // the bulk exchange may hold a pointer to `shm` (IpcBulk): let it go first
        struct Gone;
        impl BulkExchange for Gone {
            fn allgather_device(&mut self, _: *const c_void, _: *mut c_void, _: u64, _: ffi::OvmStream) -> i32 {
                1
            }
        }
        self.bulk = Box::new(Gone);
        // SAFETY: created by ovm_hip_shm_comm_create, destroyed once.
        unsafe { ffi::ovm_hip_shm_comm_destroy(self.shm) };
    }
}

impl HipDevice {
    /// `Coordinator::prove` as one joint proof over `comm.world` ranks. `ctx` holds this rank's traces
    /// (device data) -- and, for the traces of other ranks, entries with the right AIR index and height
    /// whose matrices are empty; `owners[i]` = rank of `ctx.per_trace[i]`, non-decreasing in sorted
    /// order. Returns the encoded proof on rank 0 (`Proof::decode_from_bytes` reads it), `None` elsewhere.
    pub fn prove_sharded(
        &self,
        comm: &mut Box<ShardComm>,
        transcript: &mut DuplexSpongeHip,
        mpk: &DeviceMultiStarkProvingKey<HipBackend>,
        ctx: &ProvingContext<HipBackend>,
        owners: &[u32],
    ) -> Result<Option<Vec<u8>>, HipError> {
        assert_eq!(owners.len(), ctx.per_trace.len());
        let abi = abi_traces(ctx);
        let key = mpk.per_air.first().expect("proving key without AIRs").other_data.key.0;
        let mut ends = [0u64; 5];
        HipError::from_result("ovm_hip_proof_size", unsafe {
            ffi::ovm_hip_proof_size(key, abi.traces.len() as u32, abi.traces.as_ptr(), ends.as_mut_ptr())
        })?;
        let mut out = vec![0u8; if comm.rank == 0 { ends[4] as usize } else { 0 }];
        let mut len = 0u64;
        let shard = comm.abi();
        // SAFETY: all pointers are valid for the call; `comm` outlives it.
        HipError::from_result("ovm_hip_prove_sharded", unsafe {
            ffi::ovm_hip_prove_sharded(
                key,
                &shard,
                abi.traces.len() as u32,
                abi.traces.as_ptr(),
                owners.as_ptr(),
                transcript.state_mut(),
                out.as_mut_ptr(),
                out.len() as u64,
                &mut len,
                ptr::null_mut(),
                self.stream(),
            )
        })?;
        Ok((comm.rank == 0).then(|| {
            out.truncate(len as usize);
            out
        }))
    }
}
